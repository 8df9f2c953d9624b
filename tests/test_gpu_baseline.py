"""GPU: cross-fit Whittaker baseline through the C ABI (rocco_hip_crossfit_whittaker_baseline_matrix_f64)
against the golden vectors of the reference's backend and against the CPU oracle: bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "baseline_vectors.npz")


def test_golden_baselines_bit_for_bit(gpu):
    from rocco_amd.inference import crossfit_whittaker_baseline

    gold = np.load(GOLD)
    for name in gold["names"]:
        got = crossfit_whittaker_baseline(gold[f"{name}_matrix"], float(gold[f"{name}_lambda"]))
        assert got.dtype == np.float64 and got.shape == gold[f"{name}_matrix"].shape
        assert got.tobytes() == gold[f"{name}_baseline"].tobytes(), name
    # one-dimensional input (rocco/_baseline.c:74-81)
    name = str(gold["names"][-1])
    row = gold[f"{name}_matrix"][0]
    one = crossfit_whittaker_baseline(row, float(gold[f"{name}_lambda"]))
    assert one.shape == row.shape and one.tobytes() == gold[f"{name}_baseline"][0].tobytes()
    with pytest.raises(ValueError):
        crossfit_whittaker_baseline(np.zeros((2, 2, 2)), 1.0)


@pytest.mark.parametrize("rows,cols", [(1, 25), (3, 63), (5, 64), (7, 65), (64, 129), (65, 4097), (130, 20001), (2, 300007)])
def test_random_matrices_match_oracle(gpu, oracle, rows, cols):
    from rocco_amd.inference import _consenrich_whittaker_lambda, crossfit_whittaker_baseline

    rng = np.random.default_rng(rows * 1000 + cols)
    m = rng.normal(0.0, 1.5, size=(rows, cols))
    m[rng.random(m.shape) < 0.1] = 0.0
    for block in (3, 101):
        lam = _consenrich_whittaker_lambda(min(block, cols))
        assert crossfit_whittaker_baseline(m, lam).tobytes() == oracle.crossfit_whittaker_baseline(m, lam).tobytes(), (rows, cols, block)


def test_local_background_matrix(gpu, oracle):
    from rocco_amd.inference import _estimate_local_background_matrix

    rng = np.random.default_rng(3)
    m = rng.normal(size=(4, 2000))
    base, window, lam = _estimate_local_background_matrix(m)
    assert window == 101 and lam == 7.0 * ((101.0 * 0.15915494) ** 4)
    assert base.tobytes() == oracle.crossfit_whittaker_baseline(m, lam).tobytes()
    z, w0, l0 = _estimate_local_background_matrix(rng.normal(size=(2, 24)))
    assert w0 == 0 and l0 == 0.0 and not z.any()


def test_factor_is_extended_for_longer_rows(gpu, oracle):
    """The LDL^T factor is kept per penalty and EXTENDED when a longer matrix arrives (its entries do not depend on
    the length except the last two): ascending, descending and mixed lengths must all reproduce the oracle."""
    from rocco_amd.inference import _consenrich_whittaker_lambda, crossfit_whittaker_baseline

    rng = np.random.default_rng(12)
    lam = _consenrich_whittaker_lambda(57)  # a penalty no other test uses: the factor starts from scratch here
    for cols in (25, 26, 31, 600, 599, 40000, 8, 40001, 1000, 250000, 30):
        m = rng.normal(size=(3, cols))
        assert crossfit_whittaker_baseline(m, lam).tobytes() == oracle.crossfit_whittaker_baseline(m, lam).tobytes(), cols


def test_factor_walked_on_the_device_gives_the_same_baselines(gpu, oracle, monkeypatch):
    """The factor table is walked by two host threads by default (whittaker_host.cpp) and by one GPU lane per parity with
    ROCCO_HIP_FACTOR_ON_DEVICE=1: the same IEEE operations in the same order, so the same baselines, here against the
    oracle for both, from scratch and extended (penalties no other test uses)."""
    from rocco_amd.inference import _consenrich_whittaker_lambda, crossfit_whittaker_baseline

    rng = np.random.default_rng(13)
    for where, window in (("1", 61), ("0", 63)):
        monkeypatch.setenv("ROCCO_HIP_FACTOR_ON_DEVICE", where)
        lam = _consenrich_whittaker_lambda(window)
        for cols in (27, 5000, 4999, 120001, 300):
            m = rng.normal(size=(2, cols))
            assert crossfit_whittaker_baseline(m, lam).tobytes() == oracle.crossfit_whittaker_baseline(m, lam).tobytes(), (where, cols)
