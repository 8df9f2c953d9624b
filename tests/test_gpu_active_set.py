"""Active-set rounds (frozen blocks skipped after a survey) must not change anything: benchmark-scale
chromosomes solved with the active set on, off, and by the CPU oracle give the same penalty, count
and solution bytes.  chr14 / chr20 of the benchmark genome end on the exact spine, chr21 / chr22 on
the certified path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scores(name, K=100, seed=20240):
    import torch
    from rocco_amd import rocco as rr
    from rocco_amd import synth

    genome = synth.chrom_loci(50, None)
    idx = [i for i, (nm, _) in enumerate(genome) if nm == name][0]
    n = genome[idx][1]
    m = synth.hash_matrix_device(K, n, synth.chrom_seed(seed, idx), device=torch.device("cuda:0"))
    s = torch.empty(n, dtype=torch.float64, device="cuda:0")
    rr.score_central_tendency_chrom_device(m, s)
    return s


@pytest.mark.parametrize("name", ["chr14", "chr20", "chr21", "chr22"])
def test_active_set_is_invisible(gpu, name):
    import torch
    from oracle import pyoracle
    from rocco_amd import _native, dp

    s = _scores(name)
    solver = _native.solver_for(0)
    out = {}
    try:
        for flag in (1, 0):
            solver.set("active_set", flag)
            sol, obj, det = dp.solve_chrom_exact_device(s, budget=0.02, gamma=1.0)
            torch.cuda.synchronize()
            out[flag] = (sol.cpu().numpy().copy(), obj, det)
    finally:
        solver.set("active_set", 1)
    on, off = out[1], out[0]
    assert np.array_equal(on[0], off[0])
    assert on[2]["selection_penalty"] == off[2]["selection_penalty"]
    assert on[2]["selected_count"] == off[2]["selected_count"]
    assert on[2]["_path"] == off[2]["_path"]
    # and both equal the reference algorithm on the CPU
    ref_sol, ref_obj, ref_det = pyoracle.solve_chrom_exact(s.cpu().numpy(), budget=0.02, gamma=1.0,
                                                           return_details=True)
    assert np.array_equal(on[0], ref_sol)
    assert on[2]["selected_count"] == ref_det["selected_count"]
    assert abs(on[2]["selection_penalty"] - ref_det["selection_penalty"]) <= 1e-12
    assert abs(on[1] - ref_obj) <= 1e-9 * max(1.0, abs(ref_obj))
