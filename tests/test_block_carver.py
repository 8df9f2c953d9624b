"""CPU: the first-fit carver the budget estimates use over blocks borrowed from the batch scoring
(rocco_amd.inference.BlockCarver) and the out-of-memory policy in front of the library's entry points (rocco_amd._native._Library)."""
import ctypes

import pytest


def test_carver_first_fit_and_reset():
    import torch
    from rocco_amd.inference import BlockCarver

    a, b = torch.zeros(1000, dtype=torch.float64), torch.zeros(300, dtype=torch.float64)
    carver = BlockCarver([a, b])
    assert carver.capacity() == 1300
    assert carver.fits([600, 300, 200]) and not carver.fits([600, 500, 300])  # 600 in a; 500 fits nowhere
    x = carver.take(600)
    y = carver.take(300)  # a has 1000 - 640 = 360 left (steps of 64 elements)
    z = carver.take(250)  # a has 40 left: goes to b
    assert x.data_ptr() == a.data_ptr() and y.data_ptr() == a.data_ptr() + 640 * 8 and z.data_ptr() == b.data_ptr()
    assert x.numel() == 600 and y.numel() == 300 and z.numel() == 250
    with pytest.raises(MemoryError):
        carver.take(100)
    carver.reset()
    assert carver.take(1000).data_ptr() == a.data_ptr()
    # `fits` steps as `take` does: what it promises can be taken
    carver.reset()
    assert carver.fits([1, 1, 1]) and not BlockCarver([torch.zeros(100, dtype=torch.float64)]).fits([50, 50])
    x.fill_(1.0)
    assert float(a[:600].sum()) == 600.0  # views, not copies


def test_library_wrapper_passes_results_through_and_leaves_other_functions_alone():
    from rocco_amd import _native

    class Fake:
        def __init__(self):
            self.calls = 0

            def status(value):
                self.calls += 1
                return value

            status.restype = ctypes.c_int
            self.status = status

            def text():
                return b"x"

            text.restype = ctypes.c_char_p
            self.text = text

    fake = Fake()
    lib = _native._Library(fake)
    assert lib.status(0) == 0 and lib.status(_native.EINVAL) == _native.EINVAL and fake.calls == 2
    assert lib.text is fake.text  # not a status-returning entry point: untouched
    assert lib.status is lib.status  # wrapped once
