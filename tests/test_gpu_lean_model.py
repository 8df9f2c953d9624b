"""The lean rounding-model evaluation (lean.hip: lean_model_kernel; rocco_hip_delta_model_lean_f64): whenever it
certifies a count, that count is the reference's own (rocco/_chain_dp.c through the oracle's exact restatement), on
arrays and penalties chosen to stress chunk modes, binade edges, the chain's ends and tile borders; and it certifies
almost always on realistic scores."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scores(kind, n, rng):
    if kind == "round5":
        s = np.round(rng.gamma(1.0, 0.3, n), 5)
        s[rng.integers(0, n, max(1, n // 40))] += np.round(rng.gamma(6.0, 1.0, max(1, n // 40)), 5)
        return s
    if kind == "int":
        return rng.integers(-3, 6, n).astype(float)
    if kind == "normal":
        return rng.normal(0.2, 1.0, n)
    if kind == "offset":
        return 1.0e3 + rng.gamma(1.0, 1.0, n)
    if kind == "dyadic":  # scores on a coarse binary grid: exact half-way ties of rn_u(s) become possible
        return rng.integers(0, 4096, n) / 1024.0
    return rng.normal(0.0, 1.0e6, n)


@pytest.mark.parametrize("kind", ["round5", "int", "normal", "offset", "dyadic", "huge"])
@pytest.mark.parametrize("n", [2, 33, 8191, 8193, 70001, 300000])
def test_certified_counts_are_the_references(gpu, oracle, kind, n):
    import torch

    from rocco_amd import delta

    rng = np.random.default_rng([n, len(kind)])
    s = _scores(kind, n, rng)
    gamma = float(rng.choice([0.5, 1.0, 3.0]))
    costs = oracle.build_switch_costs(s, gamma)
    s_t = torch.from_numpy(s).cuda()
    lam_ref = float(np.quantile(s, 0.9))
    sabs = float(np.max(np.abs(s)))
    margin = gamma + (float(np.max(s)) - float(np.min(s))) + sabs + 4.0
    emap = delta.delta_build_map_device(s_t, gamma, lam_ref, margin)
    spread = max(1e-9, 1e-3 * max(1.0, abs(lam_ref)))
    lams = [lam_ref] + list(lam_ref + spread * rng.uniform(-1.0, 1.0, 40)) + list(lam_ref + 1e-9 * rng.uniform(-1, 1, 23))
    got = delta.delta_model_lean_device(s_t, gamma, lams, emap)
    full = delta.delta_probe_device(s_t, gamma, lams, emap)
    n_open = 0
    for lam, (count, is_open), f in zip(lams, got, full):
        ref = oracle.solve_penalized_chain(s, costs, lam)[2]
        if not is_open:
            assert count == ref, (kind, n, lam, count, ref)
        else:
            n_open += 1
        if f["uncertain"] == 0:
            assert f["count"] == ref
    if kind in ("round5", "normal"):
        assert n_open <= 3, (kind, n, n_open)


def test_penalties_that_tie_on_the_grid_are_left_open_or_right(gpu, oracle):
    import torch

    from rocco_amd import delta

    rng = np.random.default_rng(5)
    n = 50000
    s = np.round(rng.gamma(1.0, 0.3, n), 5)
    s_t = torch.from_numpy(s).cuda()
    costs = oracle.build_switch_costs(s, 1.0)
    lam_ref = 0.75
    emap = delta.delta_build_map_device(s_t, 1.0, lam_ref, 1.0 + float(np.ptp(s)) + float(np.max(np.abs(s))) + 4.0)
    # -lambda exactly half-way between grid points of several binades
    lams = [0.75 + 2.0 ** -k for k in (30, 35, 40, 45, 48, 50)] + [0.75]
    for lam, (count, is_open) in zip(lams, delta.delta_model_lean_device(s_t, 1.0, lams, emap)):
        if not is_open:
            assert count == oracle.solve_penalized_chain(s, costs, lam)[2]
