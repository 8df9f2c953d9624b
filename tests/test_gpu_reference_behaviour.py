"""GPU: the reference's own behavioural tests of the count-path scoring (tests/test_rocco.py:262-393 of the
reference), run against this package's drop-in functions with the same inputs and the same assertions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_precision_floor_raises_standard_error(gpu):
    """tests/test_rocco.py:262-286"""
    from rocco_amd.inference import _score_centered_wls_matrix

    centered = np.array([[0.05, 1.0, 1.0, 0.05], [0.04, 1.0, 1.0, 0.04], [0.06, 1.0, 1.0, 0.06]], dtype=np.float64)
    low_scores, low = _score_centered_wls_matrix(centered, prior_df=6.0, precision_floor_ratio=0.0)
    high_scores, high = _score_centered_wls_matrix(centered, prior_df=6.0, precision_floor_ratio=0.25)
    assert np.isclose(high["precision_floor_ratio"], 0.25)
    assert np.all(high["standard_error"] >= low["standard_error"])
    assert np.all(high_scores <= low_scores)


def test_tied_large_matrix(gpu):
    """tests/test_rocco.py:331-345"""
    from rocco_amd.inference import _score_centered_wls_matrix

    centered = np.zeros((3, 250000), dtype=np.float64)
    scores, details = _score_centered_wls_matrix(centered, lower_bound_z=1.0, prior_df=5.0)
    assert scores.shape == (250000,)
    assert np.allclose(details["mean"], 0.0)
    assert np.allclose(details["z_scores"], 0.0)
    assert np.allclose(scores, -1.0)
    assert np.all(details["standard_error"] > 0.0)


def test_downweights_noisy_track_locally(gpu):
    """tests/test_rocco.py:348-376"""
    from rocco_amd.inference import _score_centered_wls_matrix

    x = np.linspace(-4.0, 4.0, 513, dtype=np.float64)
    smooth = 0.9 * np.sin(x) + 0.15 * np.cos(2.0 * x)
    noisy = smooth.copy()
    noisy_region = slice(180, 333)
    noisy[noisy_region] += 0.75 * np.where((np.arange(noisy_region.stop - noisy_region.start) % 2) == 0, 1.0, -1.0)
    centered = np.vstack([smooth, noisy])
    _, details = _score_centered_wls_matrix(centered, lower_bound_z=0.0, prior_df=6.0, spatial_window=31)
    simple_mean = centered.mean(axis=0)
    quiet_region = slice(40, 140)
    assert np.mean(np.abs(details["mean"][noisy_region] - smooth[noisy_region])) < np.mean(
        np.abs(simple_mean[noisy_region] - smooth[noisy_region]))
    assert np.mean(details["standard_error"][noisy_region]) > np.mean(details["standard_error"][quiet_region])


def test_crossfit_local_baseline_tracks_broad_background(gpu):
    """tests/test_rocco.py:379-393"""
    from rocco_amd.inference import _consenrich_crossfit_whittaker_baseline

    x = np.arange(129, dtype=np.float64)
    broad = 2.5 * np.exp(-0.5 * ((x - 64.0) / 18.0) ** 2)
    spike = 5.0 * np.exp(-0.5 * ((x - 64.0) / 2.5) ** 2)
    y = broad + spike
    baseline = _consenrich_crossfit_whittaker_baseline(y, block_size=41)
    residual = y - baseline
    shoulder_idx, peak_idx = 46, 64
    assert baseline.shape == y.shape
    assert baseline[shoulder_idx] > 0.5 * broad[shoulder_idx]
    assert residual[peak_idx] > 3.0 * max(residual[shoulder_idx], 1.0e-6)
    assert not _consenrich_crossfit_whittaker_baseline(np.ones(24)).any()
    with pytest.raises(ValueError):
        _consenrich_crossfit_whittaker_baseline(np.ones((2, 30)))
