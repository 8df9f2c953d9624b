"""GPU: the count-path scoring of several chromosomes at once (rocco_amd.inference.score_loci_wls_batch_device: pipelines of
matrices; per pipeline the Whittaker baselines of every matrix in one pair of launches, 16 chains per wavefront, and the
rolling variances of every row in one launch) against the single-matrix path and the oracle -- every track bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TRACKS = ("mean", "raw_variance", "prior_variance", "moderated_variance", "standard_error", "z_scores", "centered_matrix")


def _counts(rng, K, n):
    k = rng.integers(0, 4, size=(K, n))
    for c in rng.integers(0, max(1, n - 40), size=max(1, n // 1500)):
        k[:, c:c + int(rng.integers(4, 40))] += rng.integers(2, 7, size=(K, 1))
    return np.ldexp(1.0, k) - 1.0  # 2^k - 1: log2(count + 1) is exact, every track must agree bit for bit


@pytest.mark.parametrize("shapes", [[(3, 5000)], [(33, 9000), (1, 64), (2, 24), (40, 100), (65, 4097), (7, 30000)],
                                    [(100, 20000), (100, 8191), (100, 63), (100, 128)]])
def test_batch_equals_single_and_oracle(gpu, oracle, shapes):
    import torch
    from rocco_amd import inference

    rng = np.random.default_rng(len(shapes))
    hosts = [_counts(rng, K, n) for K, n in shapes]
    mats = [torch.from_numpy(h).to(gpu) for h in hosts]
    batch = inference.score_loci_wls_batch_device(mats, workers=3)
    assert len(batch) == len(mats)
    for i, (h, m, (scores, details)) in enumerate(zip(hosts, mats, batch)):
        single_scores, single_details = inference.score_loci_wls_device(m)
        assert torch.equal(scores, single_scores), i
        for key in TRACKS:
            assert torch.equal(details[key], single_details[key]), (i, key)
        assert details["local_baseline_window"] == single_details["local_baseline_window"]
        if i < 2 or h.size < 400000:
            o_scores, o_details = oracle.score_loci_wls(h)
            assert scores.cpu().numpy().tobytes() == o_scores.tobytes(), i
            assert details["centered_matrix"].cpu().numpy().tobytes() == np.asarray(o_details["centered_matrix"]).tobytes(), i


def test_whittaker_batch_row_groups_and_lengths(gpu, oracle):
    """Row counts around the 32-row groups, lengths around the 64-locus tiles and the 25-locus floor, one penalty."""
    import torch
    from rocco_amd import inference

    rng = np.random.default_rng(9)
    lam = inference._consenrich_whittaker_lambda(101)
    shapes = [(1, 25), (31, 63), (32, 64), (33, 65), (64, 127), (65, 128), (100, 129), (2, 24), (5, 4096), (3, 20001)]
    hosts = [rng.normal(size=s) for s in shapes]
    outs = inference.crossfit_whittaker_baseline_batch_device([torch.from_numpy(h).to(gpu) for h in hosts], lam)
    for h, o in zip(hosts, outs):
        assert o.cpu().numpy().tobytes() == oracle.crossfit_whittaker_baseline(h, lam).tobytes(), h.shape
    with pytest.raises(ValueError):
        m = torch.from_numpy(hosts[0]).to(gpu)
        inference.crossfit_whittaker_baseline_batch_device([m], lam, outs=[m])


def _seam_repairs():
    from rocco_amd import _native

    return int(_native.load().rocco_hip_whittaker_seam_repairs())


def test_rows_cut_into_segments_are_the_sequential_sweeps_bits(gpu, oracle, monkeypatch):
    """Round 5: long rows are cut into segments whose workgroups start from a warm-up (csrc/whittaker.hip).  Default
    settings on rows long enough to be cut (2 and 3 segments, lengths around the 64-locus tiles, a lone matrix and a
    batch): every baseline the oracle's bit for bit, and no seam had to be recomputed."""
    import torch
    from rocco_amd import inference

    rng = np.random.default_rng(51)
    lam = inference._consenrich_whittaker_lambda(101)
    before = _seam_repairs()
    shapes = [(3, 262144), (9, 300001), (2, 393279), (17, 262207)]
    hosts = [rng.normal(0.0, 1.3, size=s) for s in shapes]
    hosts[1][:, 1000:90000] = 0.0  # a long run of exact zeros
    hosts[2][:] = np.round(hosts[2] * 4.0) / 4.0  # a coarse grid: many ties
    outs = inference.crossfit_whittaker_baseline_batch_device([torch.from_numpy(h).to(gpu) for h in hosts], lam)
    for h, o in zip(hosts, outs):
        assert o.cpu().numpy().tobytes() == oracle.crossfit_whittaker_baseline(h, lam).tobytes(), h.shape
    one = inference.crossfit_whittaker_baseline_batch_device([torch.from_numpy(hosts[1]).to(gpu)], lam)[0]
    assert one.cpu().numpy().tobytes() == oracle.crossfit_whittaker_baseline(hosts[1], lam).tobytes()
    assert _seam_repairs() == before, "a warm-up of 131 072 loci did not reach the row's own values"


@pytest.mark.parametrize("segment,warm", [(4096, 64), (1024, 128), (8192, 2048), (20000, 20000)])
def test_seams_that_have_not_met_are_recomputed(gpu, oracle, monkeypatch, segment, warm):
    """The same with segments and warm-ups far too short (ROCCO_HIP_WHITTAKER_SEGMENT_LOCI / _WARM_LOCI): nearly every
    seam differs from its predecessor's end state and is recomputed from it -- some to the segment's end, which moves
    the next seam's truth.  Results are the oracle's bits all the same; the counter says the seams were recomputed."""
    import torch
    from rocco_amd import inference

    monkeypatch.setenv("ROCCO_HIP_WHITTAKER_SEGMENT_LOCI", str(segment))
    monkeypatch.setenv("ROCCO_HIP_WHITTAKER_WARM_LOCI", str(warm))
    rng = np.random.default_rng(segment + warm)
    shapes = [(3, 50000), (10, 33333), (1, 70001), (8, 2 * segment), (5, 2 * segment - 1), (4, 3 * segment + 65)]
    for lam in (inference._consenrich_whittaker_lambda(101), inference._consenrich_whittaker_lambda(9)):
        before = _seam_repairs()
        hosts = [rng.normal(0.0, 2.0, size=s) for s in shapes]
        hosts[0][:, ::7] = 0.0
        outs = inference.crossfit_whittaker_baseline_batch_device([torch.from_numpy(h).to(gpu) for h in hosts], lam)
        for h, o in zip(hosts, outs):
            assert o.cpu().numpy().tobytes() == oracle.crossfit_whittaker_baseline(h, lam).tobytes(), (h.shape, lam)
        if warm < 1000 and lam > 1000.0:
            assert _seam_repairs() > before
    monkeypatch.setenv("ROCCO_HIP_WHITTAKER_SEGMENT_LOCI", "0")  # rows are never cut: the round-3 launch
    h = rng.normal(size=(9, 40000))
    lam = inference._consenrich_whittaker_lambda(101)
    before = _seam_repairs()
    o = inference.crossfit_whittaker_baseline_batch_device([torch.from_numpy(h).to(gpu), torch.from_numpy(h[:2]).to(gpu)], lam)[0]
    assert o.cpu().numpy().tobytes() == oracle.crossfit_whittaker_baseline(h, lam).tobytes()
    assert _seam_repairs() == before


@pytest.mark.parametrize("segment,warm", [(None, None), (2048, 64), (4096, 4096)])
def test_residual_form_is_the_three_statements(gpu, oracle, monkeypatch, segment, warm):
    """Round 5: `crossfit_whittaker_residual_batch_device` folds rocco/inference.py:330-331 (minus the row medians) and 338
    (minus the baselines) into the sweeps.  Against the statements done one by one in NumPy around the oracle's baselines, bit for
    bit -- whole rows, rows cut into segments, segments whose seams are recomputed (the repair writes the residual too), with and
    without offsets, rows too short for a baseline; the fused and the unfused scoring of a batch agree on every track."""
    import torch
    from rocco_amd import inference

    if segment is not None:
        monkeypatch.setenv("ROCCO_HIP_WHITTAKER_SEGMENT_LOCI", str(segment))
        monkeypatch.setenv("ROCCO_HIP_WHITTAKER_WARM_LOCI", str(warm))
    rng = np.random.default_rng(77 + (segment or 0))
    lam = inference._consenrich_whittaker_lambda(101)
    shapes = [(3, 30000), (33, 9001), (1, 24), (2, 25), (40, 129), (8, 20000)]
    hosts = [rng.normal(3.0, 2.0, size=s) for s in shapes]
    offs = [np.median(h, axis=1) for h in hosts]
    offs[3] = None
    mats = [torch.from_numpy(h).to(gpu) for h in hosts]
    outs = inference.crossfit_whittaker_residual_batch_device(mats, [None if o is None else torch.from_numpy(o).to(gpu) for o in offs], lam)
    for h, f, o in zip(hosts, offs, outs):
        g = h if f is None else (h - f[:, None])
        want = g - oracle.crossfit_whittaker_baseline(g, lam)
        assert o.cpu().numpy().tobytes() == want.tobytes(), h.shape
    for m, h in zip(mats, hosts):
        assert m.cpu().numpy().tobytes() == h.tobytes()  # the inputs are untouched
    # one matrix alone, no offsets at all
    one = inference.crossfit_whittaker_residual_batch_device([mats[0]], None, lam)[0]
    assert one.cpu().numpy().tobytes() == (hosts[0] - oracle.crossfit_whittaker_baseline(hosts[0], lam)).tobytes()
    with pytest.raises(ValueError):
        inference.crossfit_whittaker_residual_batch_device([mats[0]], None, lam, outs=[mats[0]])
    # the whole scoring with and without the folded statements
    counts = [torch.from_numpy(_counts(rng, K, n)).to(gpu) for K, n in [(9, 30000), (33, 5000), (2, 24), (5, 64)]]
    fused = inference.score_loci_wls_batch_device(counts, workers=2)
    monkeypatch.setenv("ROCCO_BATCH_FUSED_RESIDUAL", "0")
    plain = inference.score_loci_wls_batch_device(counts, workers=2)
    for (fs, fd), (ps, pd) in zip(fused, plain):
        assert torch.equal(fs, ps)
        for key in TRACKS:
            assert torch.equal(fd[key], pd[key]), key
    monkeypatch.delenv("ROCCO_BATCH_FUSED_RESIDUAL")
    kept = [c.clone() for c in counts]
    over = inference.score_loci_wls_batch_device(counts, workers=2, overwrite_input=True)
    for c, k, (s_o, d_o), (fs, fd) in zip(counts, kept, over, fused):
        assert torch.equal(s_o, fs) and torch.equal(d_o["centered_matrix"], fd["centered_matrix"])
        assert d_o["centered_matrix"].data_ptr() == c.data_ptr()  # the caller's tensor holds the centred values


def test_residual_form_reports_a_baseline_that_is_not_finite(gpu):
    import torch
    from rocco_amd import inference

    lam = inference._consenrich_whittaker_lambda(101)
    m = torch.zeros((3, 5000), dtype=torch.float64, device=gpu)
    m[1, 2500] = float("inf")
    with pytest.raises(ValueError, match="non-finite"):
        inference.crossfit_whittaker_residual_batch_device([m], None, lam)
    m[1, 2500] = 1.0
    inference.crossfit_whittaker_residual_batch_device([m], None, lam)


def test_batch_errors(gpu):
    import torch
    from rocco_amd import inference

    ok = torch.ones((2, 300), dtype=torch.float64, device=gpu)
    bad = ok.clone()
    bad[1, 7] = float("nan")
    with pytest.raises(ValueError):
        inference.score_loci_wls_batch_device([ok, bad])
    with pytest.raises(ValueError):
        inference.score_loci_wls_batch_device([ok.to(torch.float32)])
    assert inference.score_loci_wls_batch_device([]) == []


def test_release_batch_workers_then_batch_again(gpu):
    """The pipelines keep their solver handles (and scratch) between calls; releasing them must leave the next call working
    and give the memory back."""
    import torch

    from rocco_amd import inference

    rng = np.random.default_rng(3)
    mats = [torch.from_numpy(np.ldexp(1.0, rng.integers(0, 8, (6, n))) - 1.0).to("cuda:0") for n in (5000, 333, 70000)]
    first = [s.clone() for s, _ in inference.score_loci_wls_batch_device(mats)]
    inference.release_batch_workers()
    assert inference._batch_workers == {}
    second = [s for s, _ in inference.score_loci_wls_batch_device(mats)]
    assert all(torch.equal(a, b) for a, b in zip(first, second))


def test_no_solver_buffer_grows_while_the_pipelines_run(gpu):
    """Growing a solver buffer is a hipMalloc + hipFree, each a device-wide synchronisation; the batch sizes every
    pipeline's buffers on the calling thread before its worker threads start (rocco_hip_count_path_reserve), also on the
    very first call of fresh pipelines."""
    import torch

    from rocco_amd import inference

    inference.release_batch_workers()
    rng = np.random.default_rng(12)
    mats = [torch.from_numpy(rng.poisson(6.0, size=(K, n)).astype(np.float64)).cuda()
            for K, n in ((5, 30000), (3, 9000), (6, 51000), (4, 4096), (5, 12345))]
    first = inference.score_loci_wls_batch_device([m.clone() for m in mats], workers=3)
    assert inference.last_batch_growths_in_flight == 0
    again = inference.score_loci_wls_batch_device([m.clone() for m in mats], workers=3)
    assert inference.last_batch_growths_in_flight == 0
    for (a, _da), (b, _db), m in zip(first, again, mats):
        single = inference.score_loci_wls_device(m.clone())[0]
        assert torch.equal(a, b) and torch.equal(a, single)


def test_a_tiny_contig_among_chromosomes_grows_nothing_in_flight(gpu):
    """A batch that mixes a contig of 25 .. 100 loci (its baseline window, and so its Whittaker penalty, is its own) with
    longer matrices: the device keeps one factor per penalty and every one of them is built before the pipelines start --
    no solver buffer grows, and no factor is rebuilt, while they run."""
    import torch
    from rocco_amd import inference

    rng = np.random.default_rng(77)
    shapes = [(6, 30000), (6, 60), (6, 21000), (6, 99), (6, 45), (6, 12000)]
    mats = [torch.from_numpy(rng.poisson(3.0, size=s).astype(np.float64)).to(gpu) for s in shapes]
    inference.score_loci_wls_batch_device([m.clone() for m in mats])
    out = inference.score_loci_wls_batch_device([m.clone() for m in mats])
    assert inference.last_batch_growths_in_flight == 0
    for m, (scores, _details) in zip(mats, out):
        one, _d = inference.score_loci_wls_device(m.clone())
        assert torch.equal(scores, one)


def test_chunks_within_a_memory_budget_give_the_same_tracks(gpu):
    """A budget that forces every pipeline to walk its matrices in several chunks (each paying its baseline and rolling
    launches again): bit for bit the results of the call that holds everything at once."""
    import torch

    from rocco_amd import inference

    rng = np.random.default_rng(21)
    shapes = ((4, 20000), (4, 18000), (3, 9000), (5, 30000), (4, 4096), (4, 15000), (2, 7000))
    mats = [torch.from_numpy(rng.poisson(5.0, size=s).astype(np.float64)).cuda() for s in shapes]
    whole = inference.score_loci_wls_batch_device([m.clone() for m in mats], workers=2, overwrite_input=True)
    largest = max(8 * k * n for k, n in shapes)
    tight = inference.score_loci_wls_batch_device([m.clone() for m in mats], workers=2, overwrite_input=True,
                                                  memory_budget_bytes=2 * 4 * largest)  # one or two matrices per chunk
    assert inference.last_batch_growths_in_flight == 0
    for (a, da), (b, db) in zip(whole, tight):
        assert torch.equal(a, b) and torch.equal(da["centered_matrix"], db["centered_matrix"])
        assert torch.equal(da["standard_error"], db["standard_error"])
