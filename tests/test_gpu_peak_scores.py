"""GPU: the per-peak arithmetic of the post-hoc scoring (rocco_amd/scores.py over peakscore.hip) against outputs of the
reference's own helpers (tests/golden/make_golden_scores.py): `_peak_signal_stat`, `EmpiricalNull.survival` and SciPy's
Benjamini-Hochberg as `score_peaks` applies it.  The signal statistic carries a log2: exact against a restatement with
the correctly rounded logarithm, and against the reference's (NumPy's log2) to the last place."""
import os

import numpy as np
import pytest

from log2_truth import log2_correctly_rounded

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scores_vectors.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_scoring_against_the_reference_helpers(gpu, gold):
    from scipy import stats

    from rocco_amd.scores import EmpiricalNull, score_peak_counts

    for name in gold["names"]:
        counts, lengths, binned = gold[f"{name}_counts"], gold[f"{name}_lengths"], gold[f"{name}_binned"]
        nulls = {int(k): EmpiricalNull(gold[f"{name}_null_{int(k)}"]) for k in gold[f"{name}_null_keys"]}
        got = score_peak_counts(counts, lengths, binned, nulls)
        # the statistic with the correctly rounded logarithm, NumPy for the percentile: bit for bit
        factor = 1000.0 / np.maximum(lengths.astype(np.int64), 1).astype(np.float64)
        logged = log2_correctly_rounded(np.maximum(counts * factor[:, None] + 1.0, 1.0))
        assert np.array_equal(got["signal"], np.percentile(logged, 75.0, axis=1)), name
        # the reference's own values: NumPy's log2 is within one ulp of the correctly rounded one
        assert np.allclose(got["signal"], gold[f"{name}_sig"], rtol=4e-16, atol=1e-15), name
        same = got["signal"] == gold[f"{name}_sig"]
        assert same.mean() > 0.99
        # survival and q-values: exact given the statistic (recomputed from this run's statistic with the host classes)
        want_p = np.array([nulls[int(b)].survival(s) for b, s in zip(binned, got["signal"])])
        assert np.array_equal(got["pvals"], want_p), name
        assert np.array_equal(got["qvals"], stats.false_discovery_control(want_p, method="bh")), name
        # ... and equal to the reference's wherever the statistic is
        assert np.array_equal(got["pvals"][same], gold[f"{name}_pvals"][same]), name
        assert np.allclose(got["qvals"], gold[f"{name}_qvals"], rtol=1e-12, atol=0.0), name
        assert got["bed6_scores"].max() <= 1000 and got["bed6_scores"].dtype.kind == "i"


def test_benjamini_hochberg_edge_cases(gpu):
    import torch
    from scipy import stats

    from rocco_amd.scores import benjamini_hochberg_device

    rng = np.random.default_rng(8)
    for m in (1, 2, 3, 1023, 1024, 1025, 70001):
        p = rng.random(m) ** 3
        p[rng.integers(0, m, size=max(1, m // 10))] = p[0]  # ties
        if m > 4:
            p[1], p[2] = 0.0, 1.0
        got = benjamini_hochberg_device(torch.from_numpy(p).to(gpu)).cpu().numpy()
        assert np.array_equal(got, np.atleast_1d(stats.false_discovery_control(p, method="bh"))), m


def test_empirical_null_and_single_peak_helper(gpu):
    from rocco_amd.scores import EmpiricalNull, _peak_signal_stat

    null = EmpiricalNull([3.0, 1.0, 2.0, 2.0])
    assert null.survival(2.0) == (4 - 1 + 1.0) / 5.0 and null.survival(10.0) == 1.0 / 5.0 and null.evaluate(2.0) == 0.75
    assert np.array_equal(null.survival(np.array([0.0, 2.5])), np.array([1.0, 2.0 / 5.0]))
    with pytest.raises(ValueError):
        EmpiricalNull([])
    vals = np.array([10.0, 0.0, 35.5, 7.25])
    want = np.percentile(log2_correctly_rounded(np.maximum(vals * (1000.0 / 250.0) + 1.0, 1.0)), 75.0)
    assert _peak_signal_stat(vals, 250) == want
