"""The parallel delta-form kernels (chain_fast.hip) against their sequential definition
(oracle/delta_oracle.c): counts, certification statistics, window differences and fill(LO) must
agree bit for bit, for any chunk / workgroup decomposition (sizes straddle 32 and 8192)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 31, 32, 33, 63, 64, 65, 1000, 8191, 8192, 8193, 16385, 50000, 300007]


def _scores(n, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "gamma":
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        k = max(1, n // 400)
        pos = rng.integers(0, n, size=k)
        for p in pos:
            s[p:p + int(rng.integers(3, 30))] += rng.gamma(6.0, 1.0)
        return np.round(s, 5)
    if kind == "flat":  # long unclamped stretches: scores hover around one level
        return np.round(0.5 + 0.01 * rng.standard_normal(n), 5)
    return rng.normal(size=n)


def _oracle_qexp(oracle, s, cmax):
    return oracle.grid_exponent(cmax, s.min(), s.max())


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kind", ["gamma", "flat", "normal"])
def test_probe_matches_sequential_definition(gpu, oracle, n, kind):
    import torch
    from rocco_amd.delta import delta_probe_device

    s = _scores(n, n + 3, kind)
    s_t = torch.from_numpy(s).to(gpu)
    lo, hi = float(s.min()) - 0.5, float(s.max()) + 0.5
    lambdas = [lo + (hi - lo) * f for f in (0.05, 0.2, 0.35, 0.5, 0.51, 0.8, 0.97)]
    for gamma in (1.0, 0.37):
        got = delta_probe_device(s_t, gamma, lambdas)
        for lam, g in zip(lambdas, got):
            _, want = oracle.delta_chain(s, gamma, lam, want_solution=False)
            eff = n + 1 if want["overflow"] else want["effect"]
            assert (g["count"], g["uncertain"], g["effect"], g["max_run"]) == (
                want["count"], want["uncertain"], eff, want["max_run"]), (n, kind, gamma, lam)


@pytest.mark.parametrize("n", [2, 33, 1000, 8193, 50000])
def test_probe_vector_costs(gpu, oracle, n):
    import torch
    from rocco_amd.delta import delta_probe_device

    rng = np.random.default_rng(n)
    s = _scores(n, n + 11, "gamma")
    costs = rng.uniform(0.2, 1.7, size=n - 1)
    s_t = torch.from_numpy(s).to(gpu)
    c_t = torch.from_numpy(costs).to(gpu)
    lambdas = [0.1, 0.3, 0.9, 2.5]
    got = delta_probe_device(s_t, c_t, lambdas)
    for lam, g in zip(lambdas, got):
        _, want = oracle.delta_chain(s, costs, lam, want_solution=False)
        assert (g["count"], g["uncertain"], g["max_run"]) == (want["count"], want["uncertain"], want["max_run"])


@pytest.mark.parametrize("n", SIZES)
def test_window_matches_sequential_definition(gpu, oracle, n):
    import torch
    from rocco_amd.delta import delta_window_device

    s = _scores(n, n + 5, "gamma")
    s_t = torch.from_numpy(s).to(gpu)
    for lam, half in ((0.3, 0.0), (0.3, 1e-9), (0.6, 1e-3), (1.5, 0.2)):
        sol_t, got = delta_window_device(s_t, 1.0, lam - half, lam + half)
        want_sol, want = oracle.delta_window(s, 1.0, lam - half, lam + half)
        assert np.array_equal(sol_t.cpu().numpy(), want_sol), (n, lam, half)
        for key in ("count_lo", "count_hi", "n_diff", "diff_adjacent", "overflow", "max_run"):
            assert got[key] == want[key], (key, n, lam, half)
        listed = min(want["n_diff"], 16)
        if want["n_diff"] <= 16:
            assert [d["locus"] for d in got["diffs"]] == [d["locus"] for d in want["diffs"][:listed]]
            for a, b in zip(got["diffs"], want["diffs"]):
                assert (a["margin_lo"], a["margin_hi"], a["run"], a["cls_lo"], a["cls_hi"]) == (
                    b["margin_lo"], b["margin_hi"], b["run"], b["cls_lo"], b["cls_hi"])


@pytest.mark.parametrize("n", [1000, 8193, 50000, 300007])
def test_map_mode_matches_sequential_definition(gpu, oracle, n):
    """With a binade map, clean chunks use the reference's own rounding grid; the kernels must
    still agree bit for bit with the sequential definition fed with the same map."""
    import torch
    from rocco_amd.delta import delta_build_map_device, delta_probe_device, delta_window_device

    rng = np.random.default_rng(n)
    s = _scores(n, n + 7, "gamma")
    s[rng.integers(0, n, size=n // 50)] += 40.0  # make the running value cross several binades
    s_t = torch.from_numpy(s).to(gpu)
    lam_ref = 0.35
    for margin in (14.0, 60.0):
        emap_t = delta_build_map_device(s_t, 1.0, lam_ref, margin)
        emap = emap_t.cpu().numpy()
        want_map = oracle.binade_map(s, 1.0, lam_ref, margin)
        # summation order differs between the device prefix sums and the sequential ones: codes may
        # differ only where a chunk sits exactly on a margin boundary
        assert np.mean(emap != want_map) < 1e-3
        assert (emap & 0x80 == 0).any(), "expected some clean chunks"
        lambdas = [lam_ref - 1e-3, lam_ref - 1e-7, lam_ref, lam_ref + 1e-9, lam_ref + 1e-3]
        got = delta_probe_device(s_t, 1.0, lambdas, emap_t)
        for lam, g in zip(lambdas, got):
            _, want = oracle.delta_chain(s, 1.0, lam, emap=emap, want_solution=False)
            eff = n + 1 if want["overflow"] else want["effect"]
            assert (g["count"], g["uncertain"], g["effect"], g["max_run"]) == (
                want["count"], want["uncertain"], eff, want["max_run"]), (n, margin, lam)
        for half in (0.0, 1e-10, 1e-6):
            sol_t, gw = delta_window_device(s_t, 1.0, lam_ref - half, lam_ref + half, emap_t)
            want_sol, want = oracle.delta_window(s, 1.0, lam_ref - half, lam_ref + half, emap=emap)
            assert np.array_equal(sol_t.cpu().numpy(), want_sol)
            for key in ("count_lo", "count_hi", "n_diff", "diff_adjacent", "overflow", "max_run"):
                assert gw[key] == want[key], (key, n, margin, half)


@pytest.mark.parametrize("n,kind", [(5000, "gamma"), (40000, "gamma"), (300007, "gamma"), (120000, "flat")])
def test_spine_reproduces_the_reference_exactly(gpu, oracle, n, kind):
    """The spine's counts and solution are the exact DP's (oracle = reference, bit-pinned), for
    penalties the tolerance model alone cannot always decide (hazard chunks, long runs)."""
    import torch
    from rocco_amd.delta import delta_build_map_device, delta_spine_device

    rng = np.random.default_rng(n)
    s = _scores(n, n + 13, kind)
    s[rng.integers(0, n, size=n // 40)] += 25.0  # several binade crossings of the running value
    s_t = torch.from_numpy(s).to(gpu)
    for gamma in (1.0, 0.37):
        costs = oracle.build_switch_costs(s, gamma)
        lam_ref = float(np.quantile(s, 0.9))
        reach = gamma + (s.max() - s.min()) + 4.0
        emap_t = delta_build_map_device(s_t, gamma, lam_ref, reach)
        lambdas = [lam_ref + d for d in (-1e-9, -1e-12, 0.0, 3e-13, 1e-10)]
        counts, sol_t = delta_spine_device(s_t, gamma, lambdas, emap_t, solution_index=2)
        for lam, c in zip(lambdas, counts):
            o_sol, o_val, o_cnt = oracle.solve_penalized_chain(s, costs, lam)
            assert c == o_cnt, (n, kind, gamma, lam)
        o_sol, _, _ = oracle.solve_penalized_chain(s, costs, lambdas[2])
        assert np.array_equal(sol_t.cpu().numpy(), o_sol)
