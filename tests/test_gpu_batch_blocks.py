"""GPU, round 5: the batch scoring's big blocks as the framework's (the sweeps' scratch handed in by the caller), kept and lent
to the budget estimates; the library's out-of-memory policy.  Results are the same bits with and without every one of these."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _counts(rng, K, n):
    k = rng.integers(0, 4, size=(K, n))
    for c in rng.integers(0, max(1, n - 40), size=max(1, n // 1500)):
        k[:, c:c + int(rng.integers(4, 40))] += rng.integers(2, 7, size=(K, 1))
    return np.ldexp(1.0, k) - 1.0


def test_residual_batch_with_the_callers_scratch(gpu):
    import torch
    from rocco_amd import inference

    rng = np.random.default_rng(5)
    lam = inference._consenrich_whittaker_lambda(101)
    mats = [torch.from_numpy(rng.normal(2.0, 1.0, size=s)).to(gpu) for s in [(7, 30000), (33, 4097), (2, 25)]]
    offs = [m.median(dim=1).values.contiguous() for m in mats]
    plain = inference.crossfit_whittaker_residual_batch_device(mats, offs, lam)
    need = inference.whittaker_batch_scratch_bytes([tuple(m.shape) for m in mats])
    assert need >= 2 * 8 * sum(int(m.numel()) for m in mats)
    scratch = torch.empty((need + 7) // 8, dtype=torch.float64, device=gpu)
    mine = inference.crossfit_whittaker_residual_batch_device(mats, offs, lam, scratch=scratch)
    for a, b in zip(plain, mine):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        inference.crossfit_whittaker_residual_batch_device(mats, offs, lam, scratch=scratch[:1000])


def test_blocks_are_kept_lent_and_given_back(gpu):
    import torch
    from rocco_amd import inference

    rng = np.random.default_rng(6)
    mats = [torch.from_numpy(_counts(rng, K, n)).to(gpu) for K, n in [(9, 30000), (33, 5000), (5, 64), (12, 20000)]]
    inference.drop_batch_blocks()
    plain = inference.score_loci_wls_batch_device(mats, workers=2)
    assert inference.borrow_batch_blocks(gpu.index) == []  # nothing is kept unless asked for
    kept = inference.score_loci_wls_batch_device(mats, workers=2, keep_blocks=True)
    lent = inference.borrow_batch_blocks(gpu.index)
    try:
        assert len(lent) == 4 and all(t.is_cuda and t.dtype == torch.float64 for t in lent)  # two pipelines: a block and a scratch each
        assert inference.borrow_batch_blocks(gpu.index) == []  # lent once
        for t in lent:
            t.fill_(float("nan"))  # the borrower may do what it likes with them
        while_lent = inference.score_loci_wls_batch_device(mats, workers=2, keep_blocks=True)  # allocates blocks of its own
    finally:
        inference.return_batch_blocks(gpu.index)
    again = inference.score_loci_wls_batch_device(mats, workers=2, keep_blocks=True)  # ... and these are the kept ones again
    inference.drop_batch_blocks()
    assert inference.borrow_batch_blocks(gpu.index) == []
    for results in (kept, while_lent, again):
        for (s0, d0), (s1, d1) in zip(plain, results):
            assert torch.equal(s0, s1) and torch.equal(d0["centered_matrix"], d1["centered_matrix"]) and torch.equal(d0["standard_error"], d1["standard_error"])


def test_estimate_in_borrowed_blocks_is_the_estimate(gpu):
    """The count branch's estimate with its draws carved out of a workspace (several at a time, no allocation) and without one:
    the same fraction and the same details, to the bit."""
    import torch
    from rocco_amd import budget, inference

    rng = np.random.default_rng(8)
    K, n = 6, 40000
    counts = torch.from_numpy(_counts(rng, K, n)).to(gpu)
    scores, details = inference.score_loci_wls_device(counts)
    centred = details["centered_matrix"]
    kwargs = dict(observed_scores=scores, dependence_lag_hint=101, num_null_draws=12, progress_label=None, num_processes=4,
                  return_details=True, multipliers="device")
    want_fraction, want_meta = budget.estimate_budget_nonnull_fraction_from_wild_bootstrap_null(centred, **kwargs)
    block_a = torch.empty(9 * K * n, dtype=torch.float64, device=gpu)
    block_b = torch.empty(3 * K * n, dtype=torch.float64, device=gpu)
    for pieces in ([block_a], [block_b, block_a], [block_b]):  # four draws at once; pieces of two sizes; room for one draw at a time
        budget.set_null_workspace(inference.BlockCarver(pieces))
        try:
            fraction, meta = budget.estimate_budget_nonnull_fraction_from_wild_bootstrap_null(centred, **kwargs)
        finally:
            budget.set_null_workspace(None)
        assert fraction == want_fraction
        assert meta["num_null_draws"] == want_meta["num_null_draws"]
        for key in want_meta:
            if isinstance(want_meta[key], float):
                assert meta[key] == want_meta[key] or (np.isnan(meta[key]) and np.isnan(want_meta[key])), key


def test_out_of_memory_from_the_library_is_retried_once_after_the_cache_is_handed_back(gpu):
    import torch
    from rocco_amd import _native

    class Fake:
        def __init__(self):
            self.answers = [_native.ENOMEM, _native.OK, _native.ENOMEM, _native.ENOMEM]

            def status():
                return self.answers.pop(0)

            status.restype = ctypes.c_int
            self.status = status

    held = torch.empty(1 << 26, dtype=torch.float64, device=gpu)
    del held  # (cached by the allocator now)
    assert torch.cuda.memory_reserved(gpu) > torch.cuda.memory_allocated(gpu)
    fake = Fake()
    lib = _native._Library(fake)
    assert lib.status() == _native.OK and len(fake.answers) == 2  # failed, cache handed back, called again
    assert lib.status() == _native.ENOMEM and fake.answers == []  # once more only


def test_log_scaled_input_through_the_folded_sweeps_and_rows_per_workgroup_of_the_rolling_launch(gpu, oracle):
    """`input_scale="log2p1"` (the matrix is log-scaled already: only the medians and the baselines are subtracted) through the
    batch, against the single-matrix call and the oracle; and a solver that asks for at least 4 / 8 rows per workgroup of its rolling
    launches (`rolling_group_min`) gets the same variances and the same scores as one that does not."""
    import torch
    from rocco_amd import _native, inference

    rng = np.random.default_rng(11)
    counts = [_counts(rng, K, n) for K, n in [(7, 20000), (20, 4097), (3, 64)]]  # 2^k - 1: their log2(count + 1) is exact
    hosts = [np.log2(c + 1.0) for c in counts]
    mats = [torch.from_numpy(h).to(gpu) for h in hosts]
    batch = inference.score_loci_wls_batch_device(mats, input_scale="log2p1", workers=2)
    for c, m, (scores, details) in zip(counts, mats, batch):
        one_scores, one_details = inference.score_loci_wls_device(m, input_scale="log2p1")
        assert torch.equal(scores, one_scores) and torch.equal(details["centered_matrix"], one_details["centered_matrix"])
        o_scores, o_details = oracle.score_loci_wls(c)  # (from the counts: the same numbers once their logarithms are exact)
        assert scores.cpu().numpy().tobytes() == o_scores.tobytes()
        assert details["centered_matrix"].cpu().numpy().tobytes() == np.asarray(o_details["centered_matrix"]).tobytes()
    centred = batch[0][1]["centered_matrix"]
    solver = _native.solver_for(gpu.index)
    plain = inference.wls_rolling_variances_batch_device([centred, batch[1][1]["centered_matrix"]], spatial_window=31)
    plain_scores = inference.score_centered_wls_device(centred)[0]
    try:
        for least in (4, 8):
            solver.set("rolling_group_min", least)
            grouped = inference.wls_rolling_variances_batch_device([centred, batch[1][1]["centered_matrix"]], spatial_window=31)
            assert all(torch.equal(a, b) for a, b in zip(plain, grouped))
            assert torch.equal(inference.score_centered_wls_device(centred)[0], plain_scores)
        with pytest.raises(ValueError):
            solver.set("rolling_group_min", 3)
    finally:
        solver.set("rolling_group_min", 1)
