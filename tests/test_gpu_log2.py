"""GPU: the count path's log scale, log2(max(x, 0) + 1) (rocco/inference.py:40-47), is correctly rounded on the device:
exhaustively over the integer counts 0 .. 2^24, on fractional (scale-factor multiples) and large values, against an
independent long-double / decimal evaluation.  NumPy's own log2 -- what the reference calls -- is NOT correctly rounded
and differs between its SVML and libm builds; how far this host's NumPy is from the correctly rounded value is
measured and bounded here (one ulp, a fraction of a per cent of the values)."""
import numpy as np
import pytest

from log2_truth import log2_correctly_rounded

pytestmark = pytest.mark.gpu


def _device_log2p1(x):
    import torch
    from rocco_amd.inference import log_scale_device

    return log_scale_device(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to("cuda:0")).cpu().numpy()


def test_integer_counts_exhaustive(gpu):
    counts = np.arange(0, 2 ** 24 + 1, dtype=np.float64)
    got = _device_log2p1(counts)
    want = log2_correctly_rounded(counts + 1.0)
    assert np.array_equal(got, want)
    # exact at powers of two
    k = np.arange(0, 25)
    assert np.array_equal(got[(2 ** k - 1).astype(np.int64)], k.astype(np.float64))
    # this host's NumPy against the correctly rounded value: never more than one ulp, rarely off at all
    host = np.log2(counts + 1.0)
    off = host != want
    assert np.all(np.abs(host[off] - want[off]) <= np.spacing(np.abs(want[off])) * 1.0000001)
    assert off.mean() < 2.0e-3


def test_fractional_and_large_values(gpu):
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.gamma(2.0, 2.0, size=400000) * rng.random(400000),       # scale-factor multiples of counts
                        rng.integers(0, 2 ** 40, size=100000).astype(np.float64),   # large counts
                        np.ldexp(rng.random(100000), rng.integers(-60, 60, size=100000)),
                        np.array([0.0, 0.5, 1.0, 2.0 ** 52, 1.0e300, 5.0e-324, 2.2250738585072014e-308])])
    got = _device_log2p1(x)
    want = log2_correctly_rounded(np.maximum(x, 0.0) + 1.0)
    assert np.array_equal(got, want)
    assert np.array_equal(_device_log2p1(np.array([-3.0, -0.0])), np.array([0.0, 0.0]))  # clipped at zero first


def test_other_pseudocounts_and_non_finite(gpu):
    import torch
    from rocco_amd.inference import log_scale_device

    rng = np.random.default_rng(13)
    x = rng.gamma(1.0, 3.0, size=200000)
    for pc in (0.5, 1.0e-3, 8.0):
        got = log_scale_device(torch.from_numpy(x).to(gpu), pseudocount=pc).cpu().numpy()
        assert np.array_equal(got, log2_correctly_rounded(x + pc)), pc
    bad = x.copy()
    bad[17] = np.nan
    with pytest.raises(ValueError):
        log_scale_device(torch.from_numpy(bad).to(gpu))


@pytest.mark.parametrize("family,first,count", [(0, 0, 1 << 31), (1, 0, 1 << 30), (2, 1, 1 << 32), (3, 0, 1 << 30), (4, 0, 1 << 31)])
def test_two_stage_log2_agrees_with_the_full_evaluation(gpu, family, first, count):
    """The device's log2 accepts a fast double-double stage when its result survives its error bound and falls back to the
    full evaluation (error ~2^-100) otherwise: both against the full evaluation alone on 2^30 .. 2^32 inputs per family --
    any positive finite bit pattern, values next to 1 (the only small results), every integer up to 2^32 (counts + 1),
    mantissas next to the table's cell boundaries and centres, uniform values -- with no differing result."""
    import ctypes

    from rocco_amd import _native

    solver = _native.solver_for(0)
    total = ctypes.c_uint64(0)
    done = 0
    while done < count:  # (launches of at most 2^30 inputs)
        step = min(count - done, 1 << 30)
        out = ctypes.c_uint64(0)
        _native.check(_native.load().rocco_hip_log2_selfcheck(solver.handle, family, 20240 + done, first + done, step, ctypes.byref(out), None),
                      "rocco_hip_log2_selfcheck")
        total.value += out.value
        done += step
    assert total.value == 0
