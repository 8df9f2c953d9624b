"""GPU: the binade map built by the lean kernels (lean.hip: lean_map_kernel + lean_mapcode_kernel; what the budgeted solve
uses for the first map of its compacted problems) against the map of the general kernels (chain_fast.hip K1 ... K6), which
tests/test_gpu_delta_kernels.py holds against oracle/delta_oracle.c: the same per-locus operations in the same order and the
same reduction trees, so the same bytes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tracks(rng, n, kind):
    if kind == "peaks":
        s = np.round(rng.gamma(1.0, 0.3, n), 5)
        at = rng.integers(0, n, max(1, n // 40))
        s[at] += np.round(rng.gamma(6.0, 1.0, at.size), 5)
        return s
    if kind == "normal":
        return rng.normal(0.0, 1.0, n)
    if kind == "integers":
        return rng.integers(-3, 9, n).astype(np.float64)
    if kind == "offset":
        return 1000.0 + rng.gamma(1.0, 1.0, n)
    # "level": what a compacted level looks like -- runs of kept loci between separators far below every score
    s = np.round(rng.gamma(2.0, 0.5, n), 5)
    s[rng.random(n) < 0.08] = -7.0
    return s


@pytest.mark.parametrize("kind", ["peaks", "normal", "integers", "offset", "level"])
@pytest.mark.parametrize("n", [2, 31, 32, 33, 8191, 8192, 8193, 16384, 70001, 8192 * 70 + 5, 8192 * 130])
def test_lean_map_gives_the_general_kernels_codes(gpu, kind, n):
    import torch

    from rocco_amd import delta

    rng = np.random.default_rng(n * 7 + sum(map(ord, kind)))
    s = _tracks(rng, n, kind)
    s_t = torch.from_numpy(s).cuda()
    for gamma in (1.0, 0.37):
        for q in (0.5, 0.9, 0.99):
            lam = float(np.quantile(s, q)) + 1e-3 * float(rng.uniform(-1, 1))
            margin = 1.0 + float(np.ptp(s)) + float(np.max(np.abs(s))) + 4.0 * gamma
            want = delta.delta_build_map_device(s_t, gamma, lam, margin).cpu().numpy()
            got = delta.delta_build_map_lean_device(s_t, gamma, lam, margin).cpu().numpy()
            assert np.array_equal(got, want), (kind, n, gamma, q, int((got != want).sum()), np.flatnonzero(got != want)[:5])
        # a small margin: clean chunks appear
        lam = float(np.quantile(s, 0.9))
        want = delta.delta_build_map_device(s_t, 1.0, lam, 0.25).cpu().numpy()
        got = delta.delta_build_map_lean_device(s_t, 1.0, lam, 0.25).cpu().numpy()
        assert np.array_equal(got, want), (kind, n, "small margin", int((got != want).sum()))
