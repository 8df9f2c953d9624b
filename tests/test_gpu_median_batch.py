"""rocco_hip_score_median_batch: several matrices in one launch give what one launch each gives (= np.median)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,dtype", [(2, np.float64), (3, np.float64), (10, np.float32), (50, np.float64), (100, np.float64), (101, np.float64)])
def test_batch_equals_numpy(gpu, K, dtype):
    import torch
    from rocco_amd.rocco import score_central_tendency_chrom_batch_device

    rng = np.random.default_rng(K)
    sizes = [1, 255, 256, 257, 1000, 4099, 70001, 0, 513]
    mats = [np.round(rng.gamma(1.0, 0.3, size=(K, n)), 5).astype(dtype) for n in sizes]
    mats[3][:, 5] = np.nan  # a NaN column gives NaN
    outs = score_central_tendency_chrom_batch_device([torch.from_numpy(m).to(gpu) for m in mats])
    assert len(outs) == len(mats)
    for m, o in zip(mats, outs):
        want = np.median(m.astype(np.float64), axis=0) if m.shape[1] else np.zeros(0)
        got = o.cpu().numpy()
        assert got.dtype == np.float64 and got.shape == (m.shape[1],)
        assert np.array_equal(got, want, equal_nan=True)


def test_batch_with_mixed_shapes_falls_back(gpu):
    import torch
    from rocco_amd.rocco import score_central_tendency_chrom_batch_device

    rng = np.random.default_rng(5)
    mats = [rng.normal(size=(4, 100)), rng.normal(size=(7, 300))]
    outs = score_central_tendency_chrom_batch_device([torch.from_numpy(m).to(gpu) for m in mats])
    for m, o in zip(mats, outs):
        assert np.array_equal(o.cpu().numpy(), np.median(m, axis=0))


def test_decode_batch_equals_single_decodes(gpu):
    """rocco_hip_decode_runs_batch: the runs of several solutions in three launches are those of one decode each
    (including a capacity that is too small at first, an empty and a one-locus solution)."""
    import torch
    from rocco_amd.rocco import decode_runs_batch_device, decode_runs_device

    rng = np.random.default_rng(3)
    sols = []
    for n in (1, 2, 15, 16, 17, 4095, 4096, 4097, 100003, 0):
        z = (rng.random(n) < 0.3).astype(np.uint8)
        sols.append(torch.from_numpy(z).to(gpu))
    sols = [s for s in sols if s.numel() > 0]
    got = decode_runs_batch_device(sols, capacities=[4] * len(sols))
    for s, (b, e) in zip(sols, got):
        wb, we = decode_runs_device(s)
        assert torch.equal(b, wb) and torch.equal(e, we)
        z = s.cpu().numpy()[:-1].astype(np.int8)  # the last locus is never emitted (rocco/rocco.py:180)
        d = np.diff(np.concatenate([[0], z, [0]]))
        assert np.array_equal(b.cpu().numpy(), np.flatnonzero(d == 1)) and np.array_equal(e.cpu().numpy(), np.flatnonzero(d == -1))
