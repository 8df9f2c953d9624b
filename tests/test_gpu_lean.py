"""The lean exact-arithmetic evaluation (rocco_amd/csrc/lean.hip) against the sequential definition
(oracle/delta_oracle.c without a map = the "bound" evaluation): counts bit for bit on the caller's array
(level 0) and on compacted levels, for sizes that straddle the 32-locus chunk and the 8192-locus tile;
then the whole calibration with compaction on / off against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [2, 31, 32, 33, 64, 1000, 8191, 8192, 8193, 16384, 16385, 50000, 300007]


def _scores(n, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "gamma":
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        for p in rng.integers(0, n, size=max(1, n // 400)):
            s[p:p + int(rng.integers(3, 30))] += rng.gamma(6.0, 1.0)
        return np.round(s, 5)
    if kind == "flat":  # long unclamped stretches: tiles whose function is not constant
        return np.round(0.5 + 0.01 * rng.standard_normal(n), 5)
    if kind == "ints":  # exact ties everywhere
        return rng.integers(0, 6, size=n).astype(np.float64)
    if kind == "offset":
        return np.round(rng.gamma(1.0, 0.3, size=n), 5) * 1e4 + 3e6
    return rng.normal(size=n)


def _oracle_counts(oracle, s, gamma, lams):
    return [oracle.delta_chain(s, gamma, lam, want_solution=False)[1]["count"] for lam in lams]


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kind", ["gamma", "flat", "ints", "normal", "offset"])
def test_level0_counts_match_sequential_definition(gpu, oracle, n, kind):
    import torch
    from rocco_amd.delta import delta_bound_rounds_device

    s = _scores(n, 7 * n + 1, kind)
    s_t = torch.from_numpy(s).to(gpu)
    lo, hi = float(s.min()) - 0.5, float(s.max()) + 0.5
    for gamma in ((1.0, 0.37, 7.0) if kind != "offset" else (1.0e4,)):
        # descending penalties: no round can reuse the round before it, every one runs on the caller's array
        rounds = [[lo + (hi - lo) * f for f in fr] for fr in ((0.97, 0.8, 0.51), (0.5, 0.35, 0.2, 0.05, 0.01), (0.0,))]
        got = delta_bound_rounds_device(s_t, gamma, rounds)
        for used, counts, level_len in got:
            assert level_len == n
            assert counts == _oracle_counts(oracle, s, gamma, used), (n, kind, gamma, used)


@pytest.mark.parametrize("n", [33, 1000, 8193, 50000, 300007, 1200000])
@pytest.mark.parametrize("kind", ["gamma", "flat", "ints", "normal"])
def test_compacted_levels_give_the_same_counts(gpu, oracle, n, kind):
    import torch
    from rocco_amd.delta import delta_bound_rounds_device

    s = _scores(n, 3 * n + 5, kind)
    s_t = torch.from_numpy(s).to(gpu)
    gamma = 1.0
    qs = np.quantile(s, [0.5, 0.8, 0.9, 0.95, 0.98, 0.995, 0.999])
    # ascending rounds: each can run on the loci selected at a penalty of the round before
    rounds = [[float(qs[0]), float(qs[1])], [float(qs[1]) + 1e-3, float(qs[2]), float(qs[3])],
              [float(qs[3]), float(qs[4]), float(qs[5])] + [float(qs[5]) + k * 1e-4 for k in range(1, 20)],
              [float(qs[5]) + 1e-3, float(qs[6]), float(s.max()) + 0.5]]
    got = delta_bound_rounds_device(s_t, gamma, rounds)
    lens = [g[2] for g in got]
    assert lens[0] == n and all(a >= b for a, b in zip(lens, lens[1:]))
    for used, counts, _ in got:
        assert counts == _oracle_counts(oracle, s, gamma, used), (n, kind, used, lens)
    if n >= 50000 and kind == "gamma":
        assert lens[-1] < n // 4  # the later rounds really ran on compacted levels


@pytest.mark.parametrize("seed", range(6))
def test_calibration_with_and_without_compaction(gpu, oracle, seed, monkeypatch):
    import torch
    from rocco_amd import dp

    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([3000, 20000, 70000, 400000]))
    kind = ["gamma", "flat", "ints", "normal", "gamma", "offset"][seed]
    s = _scores(n, seed, kind)
    gamma = float(rng.choice([0.5, 1.0, 3.0])) * (1.0e4 if kind == "offset" else 1.0)
    budget = float(rng.choice([0.005, 0.02, 0.1]))
    target = int(np.floor(n * budget))
    ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, gamma), target)
    s_t = torch.from_numpy(s).to(gpu)
    for lean, compact in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("ROCCO_HIP_LEAN", lean)
        monkeypatch.setenv("ROCCO_HIP_COMPACT", compact)
        (pen, sol_t, val, cnt, info), = dp.calibrate_batch_device([s_t], [gamma], [target])
        assert pen == ref[0] and cnt == ref[3], (kind, n, gamma, budget, lean, compact, info)
        assert np.array_equal(sol_t.cpu().numpy(), ref[1]), (kind, n, lean, compact, info)
        assert abs(val - ref[2]) <= 1e-9 * max(1.0, abs(ref[2]))


@pytest.mark.parametrize("model,fills", [("0", "0"), ("1", "1"), ("0", "1")])
def test_lean_model_kernel_and_scratch_fills_switched(gpu, oracle, monkeypatch, model, fills):
    """ROCCO_HIP_LEAN_MODEL=0 sends every rounding-model probe through the full kernels, ROCCO_HIP_LEAN_FILLS=1 fills
    the lean rounds' scratch before every round instead of trusting the finish kernel to restore it: same results."""
    import torch

    from rocco_amd import dp, synth
    from rocco_amd.rocco import score_central_tendency_chrom_device

    ns = [700000, 150000, 42000]
    scores = [score_central_tendency_chrom_device(synth.hash_matrix_device(8, n, seed=977 + i)) for i, n in enumerate(ns)]
    targets = [int(np.floor(n * 0.02)) for n in ns]
    base = dp.calibrate_batch_device(scores, [1.0] * 3, targets)
    monkeypatch.setenv("ROCCO_HIP_LEAN_MODEL", model)
    monkeypatch.setenv("ROCCO_HIP_LEAN_FILLS", fills)
    other = dp.calibrate_batch_device(scores, [1.0] * 3, targets)
    for s_t, target, a, b in zip(scores, targets, base, other):
        assert a[0] == b[0] and a[3] == b[3] and torch.equal(a[1], b[1])
        s = s_t.cpu().numpy()
        ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, 1.0), target)
        assert b[0] == ref[0] and b[3] == ref[3] and np.array_equal(b[1].cpu().numpy(), ref[1])
