"""GPU: the threshold search sequenced by the device (rocco_amd/csrc/chain.hip: director kernel + chained lean launches)
against the oracle's calibration (rocco/dp.py:89-164 restated) and against the host-sequenced search.  By default only
genome-sized batches take the chain; ROCCO_HIP_CHAIN=1 forces it, =0 forbids it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model_chain_counters():
    import ctypes

    from rocco_amd import _native

    out = (ctypes.c_longlong * 4)()
    _native.load().rocco_hip_model_chain_counters(out)
    return list(out)  # chains, counts taken over, counts answered from them, counts asked for that a chain had not evaluated


def _written_counters():
    import ctypes

    from rocco_amd import _native

    out = (ctypes.c_longlong * 2)()
    _native.load().rocco_hip_model_chain_written_counters(out)
    return list(out)  # solutions the chains wrote themselves, final windows answered from them


def _solve(scores_list, gammas, targets, chain, monkeypatch):
    import torch

    from rocco_amd import dp

    monkeypatch.setenv("ROCCO_HIP_CHAIN", "1" if chain else "0")
    tensors = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scores_list]
    return dp.calibrate_batch_device(tensors, gammas, targets)


def _tracks(rng, n, kind):
    if kind == "peaks":  # a noise floor with enriched stretches: the shape of the benchmark's tracks
        s = np.round(rng.gamma(1.0, 0.3, n), 5)
        for p in range(50, max(51, n - 50), 1500):
            s[p:p + int(rng.integers(4, 40))] += rng.gamma(6.0, 1.0)
        return s
    if kind == "normal":
        return rng.normal(0.0, 1.0, n)
    if kind == "integers":  # ties everywhere
        return rng.integers(-3, 9, n).astype(np.float64)
    if kind == "offset":  # far from zero: the rounding model's magnitudes
        return 1000.0 + rng.gamma(1.0, 1.0, n)
    return np.round(rng.normal(0.2, 0.05, n), 3)  # "flat": a steep cliff in the count


@pytest.mark.parametrize("kind", ["peaks", "normal", "integers", "offset", "flat"])
def test_chained_search_gives_the_oracles_calibration(gpu, oracle, monkeypatch, kind):
    rng = np.random.default_rng(sum(map(ord, kind)))
    sizes = [8191, 8192, 8193, 70000, 3, 2, 262145]
    scores = [_tracks(rng, n, kind) for n in sizes]
    gammas = [1.0, 0.5, 2.0, 1.0, 1.0, 1.0, 3.0]
    targets = [int(np.floor(n * b)) for n, b in zip(sizes, (0.02, 0.1, 0.005, 0.03, 0.5, 0.0, 0.02))]
    chained = _solve(scores, gammas, targets, True, monkeypatch)
    plain = _solve(scores, gammas, targets, False, monkeypatch)
    for s, gamma, target, a, b in zip(scores, gammas, targets, chained, plain):
        ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, gamma), target)
        assert a[0] == ref[0] and a[3] == ref[3], (kind, len(s), a[0], ref[0])
        assert np.array_equal(a[1].cpu().numpy(), ref[1])
        assert b[0] == ref[0] and b[3] == ref[3]


def test_chained_search_with_pilot_on_a_long_chromosome(gpu, oracle, monkeypatch):
    """Long enough for the sampled pilot (128 tiles and more), beside short ones that search without it; the chain must
    finish the search itself (passes: chain rounds + binade map + rounding-model rounds + window)."""
    rng = np.random.default_rng(77)
    sizes = [1_300_000, 40_000, 1_100_000]
    scores = [_tracks(rng, n, "peaks") for n in sizes]
    targets = [int(np.floor(n * 0.02)) for n in sizes]
    chained = _solve(scores, [1.0] * 3, targets, True, monkeypatch)
    for s, target, a in zip(scores, targets, chained):
        ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, 1.0), target)
        assert a[0] == ref[0] and a[3] == ref[3]
        assert np.array_equal(a[1].cpu().numpy(), ref[1])
        assert a[4]["path"] == 1 and a[4]["passes"] <= 24


def test_chain_leaves_cost_vectors_and_tiny_problems_to_the_host(gpu, oracle, monkeypatch):
    """A batch in which nothing is eligible (switch-cost vectors, single loci) and a mixed one."""
    import torch

    from rocco_amd import dp

    monkeypatch.setenv("ROCCO_HIP_CHAIN", "1")
    rng = np.random.default_rng(5)
    s = rng.normal(0.0, 1.0, 30000)
    costs = rng.uniform(0.5, 2.0, s.size - 1)
    out = dp.calibrate_batch_device([torch.from_numpy(s).cuda(), torch.from_numpy(s[:1].copy()).cuda(), torch.from_numpy(s).cuda()],
                                    [torch.from_numpy(costs).cuda(), 1.0, 1.5], [600, 0, 900])
    ref0 = oracle.calibrate_selection_penalty(s, costs, 600)
    ref2 = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, 1.5), 900)
    assert out[0][0] == ref0[0] and out[0][3] == ref0[3] and np.array_equal(out[0][1].cpu().numpy(), ref0[1])
    assert out[2][0] == ref2[0] and out[2][3] == ref2[3] and np.array_equal(out[2][1].cpu().numpy(), ref2[1])


@pytest.mark.parametrize("kind", ["peaks", "normal", "integers", "offset", "flat"])
@pytest.mark.parametrize("model_chain,follow", [("1", "1"), ("0", "1"), ("1", "0")])
def test_chained_rounding_model_rounds_give_the_oracles_calibration(gpu, oracle, monkeypatch, kind, model_chain, follow):
    """The last bisection steps walked by the device (model_chain.hip: the host follows its facts through host-coherent
    memory and replays the reference's steps from them) against the oracle, with the threshold search chained or
    sequenced by the host, the report followed or copied at the end of the stream."""
    monkeypatch.setenv("ROCCO_HIP_MODEL_CHAIN", model_chain)
    monkeypatch.setenv("ROCCO_HIP_CHAIN_FOLLOW", follow)
    rng = np.random.default_rng(1000 + sum(map(ord, kind)))
    sizes = [8191, 8193, 70000, 3, 262145, 500000]
    scores = [_tracks(rng, n, kind) for n in sizes]
    gammas = [1.0, 0.5, 2.0, 1.0, 3.0, 1.0]
    targets = [int(np.floor(n * b)) for n, b in zip(sizes, (0.02, 0.1, 0.005, 0.5, 0.02, 0.03))]
    before = _model_chain_counters()
    for chain in (True, False):
        out = _solve(scores, gammas, targets, chain, monkeypatch)
        for s, gamma, target, a in zip(scores, gammas, targets, out):
            ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, gamma), target)
            assert a[0] == ref[0] and a[3] == ref[3], (kind, chain, len(s), a[0], ref[0])
            assert np.array_equal(a[1].cpu().numpy(), ref[1])
    if model_chain == "0":
        assert _model_chain_counters() == before


@pytest.mark.parametrize("kind", ["peaks", "normal", "offset"])
def test_the_director_asks_what_the_hosts_replay_asks(gpu, oracle, monkeypatch, kind):
    """A batch whose every problem ends on a compacted level: the rounding-model rounds run as a chain, and every count the
    host's replay of the reference's bisection asks for is among the counts the chain evaluated (the director mirrors
    search.cpp's walk; a divergence would still be correct -- regular rounds answer -- but slow, and show here)."""
    rng = np.random.default_rng(2000 + sum(map(ord, kind)))
    sizes = [300_000, 120_000, 90_000, 500_000]
    scores = [_tracks(rng, n, kind) for n in sizes]
    targets = [int(np.floor(n * 0.02)) for n in sizes]
    for chain in (True, False):
        before = _model_chain_counters()
        out = _solve(scores, [1.0] * len(sizes), targets, chain, monkeypatch)
        after = _model_chain_counters()
        for s, target, a in zip(scores, targets, out):
            ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, 1.0), target)
            assert a[0] == ref[0] and a[3] == ref[3]
            assert np.array_equal(a[1].cpu().numpy(), ref[1])
        assert after[0] > before[0] and after[2] > before[2], (kind, chain, before, after)
        assert after[3] == before[3], (kind, chain, before, after)


@pytest.mark.parametrize("write", ["1", "0"])
def test_solutions_written_by_the_chain_are_the_oracles(gpu, oracle, monkeypatch, write):
    """The window that ends a calibration (certify + write the solution of the final penalty) is answered from what the
    chain of rounding-model rounds left: the class words of its certified evaluation at that penalty, turned into bytes by
    lean_write_solutions_kernel.  ROCCO_HIP_CHAIN_WRITE=0: the windows run.  Solutions, penalties, counts and objective
    values against the oracle either way."""
    monkeypatch.setenv("ROCCO_HIP_CHAIN_WRITE", write)
    rng = np.random.default_rng(777)
    sizes = [310_000, 95_000, 140_000]
    scores = [_tracks(rng, n, "peaks") for n in sizes]
    targets = [int(np.floor(n * 0.02)) for n in sizes]
    before = _written_counters()
    out = _solve(scores, [1.0] * 3, targets, False, monkeypatch)
    after = _written_counters()
    for s, target, a in zip(scores, targets, out):
        costs = oracle.build_switch_costs(s, 1.0)
        ref = oracle.calibrate_selection_penalty(s, costs, target)
        assert a[0] == ref[0] and a[3] == ref[3]
        assert np.array_equal(a[1].cpu().numpy(), ref[1])
        assert np.isclose(a[2], ref[2], rtol=1e-12, atol=1e-9)  # penalised value (summation order differs)
    if write == "1":
        # (a bisection whose final upper end was decided without an evaluation of the chain keeps its window)
        assert after[0] - before[0] >= 1 and after[1] - before[1] == after[0] - before[0], (before, after)
    else:
        assert after == before


def test_model_chain_depth_override_and_single_problem(gpu, oracle, monkeypatch):
    """One compacted problem alone (deep trees: up to 63 penalties a round) and a forced depth."""
    rng = np.random.default_rng(4242)
    s = _tracks(rng, 300_000, "peaks")
    ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, 1.0), 6000)
    for depth in (None, "2", "5", "6"):
        if depth is None:
            monkeypatch.delenv("ROCCO_HIP_MODEL_DEPTH", raising=False)
        else:
            monkeypatch.setenv("ROCCO_HIP_MODEL_DEPTH", depth)
        for chain in (True, False):
            a = _solve([s], [1.0], [6000], chain, monkeypatch)[0]
            assert a[0] == ref[0] and a[3] == ref[3], (depth, chain)
            assert np.array_equal(a[1].cpu().numpy(), ref[1])


@pytest.mark.parametrize("mode", ["1", "2", "3", "4"])
def test_rounds_as_one_launch_give_the_same_calibration(gpu, oracle, monkeypatch, mode):
    """Round 5's experiment, kept as an option (ROCCO_HIP_CHAIN_FUSED, default 0; csrc/lean.hip: lean_round_chain_kernel): a
    chained round's compactions, evaluation and finish as ONE launch -- blocks and tiles by tickets, the workgroup that
    completes a (chromosome, penalty) pair's last tile finishes the pair, the next director restores tickets and counters.
    Measured slower than three launches, but it must give the same penalties, counts and
    solutions: a batch large enough for both chains, against the three-launch form and (one chromosome) the oracle."""
    import torch

    from rocco_amd import dp, synth
    from rocco_amd import rocco as rr

    monkeypatch.setenv("ROCCO_HIP_CHAIN_MIN_TILES", "1")
    sizes = [1_200_001, 700_000, 2_100_000, 65_000, 900_123, 1_500_000]
    scores = [rr.score_central_tendency_chrom_device(synth.hash_matrix_device(6, n, 900 + i)) for i, n in enumerate(sizes)]
    targets = [int(np.floor(n * 0.02)) for n in sizes]
    monkeypatch.setenv("ROCCO_HIP_CHAIN_FUSED", "0")
    want = dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
    monkeypatch.setenv("ROCCO_HIP_CHAIN_FUSED", mode)
    for _rep in range(2):  # (the second call starts from the scratch the first one left)
        got = dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
        for g, w in zip(got, want):
            assert g[0] == w[0] and g[3] == w[3] and g[4]["path"] == w[4]["path"] and torch.equal(g[1], w[1])
    o_sol, _obj, o_det = oracle.solve_chrom_exact(scores[3].cpu().numpy(), budget=0.02, gamma=1.0, return_details=True)
    assert got[3][0] == o_det["selection_penalty"] and np.array_equal(got[3][1].cpu().numpy(), o_sol)
