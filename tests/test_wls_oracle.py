"""CPU: the oracle's restatement of the centred-WLS backend (oracle/wls_oracle.c) against the golden
vectors written from the reference's own backend, and against that backend itself when its build
(oracle/_ref/libwls_ref.so) is present."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "wls_vectors.npz")


def run_case(oracle, gold, name):
    lbz, pdf, me, use_me, win, pfr = gold[f"{name}_params"]
    res = oracle.score_centered_wls(gold[f"{name}_matrix"], lower_bound_z=lbz, prior_df=pdf,
                                    min_effect=(me if use_me else None), spatial_window=int(win),
                                    precision_floor_ratio=pfr)
    scores, mean, raw, prior, mod, se, df, window = res
    return np.stack([mean, raw, prior, mod, se, scores]), np.array([df, window])


def test_oracle_reproduces_golden_wls_tracks_bit_for_bit(oracle):
    gold = np.load(GOLD)
    for name in gold["names"]:
        tracks, dfw = run_case(oracle, gold, name)
        assert tracks.tobytes() == gold[f"{name}_tracks"].tobytes(), name
        assert np.array_equal(dfw, gold[f"{name}_df_window"]), name


def test_oracle_matches_compiled_reference_backend(oracle):
    path = os.path.join(ROOT, "oracle", "_ref", "libwls_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libwls_ref.so not built (reference not present)")
    dp = ctypes.POINTER(ctypes.c_double)
    ref = ctypes.CDLL(path).rocco_score_centered_wls_f64
    ref.argtypes = [dp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                    ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, dp, dp, dp, dp, dp, dp,
                    ctypes.POINTER(ctypes.c_int)]
    ref.restype = ctypes.c_int
    rng = np.random.default_rng(11)
    for n in (1, 4, 5, 33, 1000, 50001):
        for K in (1, 4):
            m = np.ascontiguousarray(np.round(rng.normal(0, 1, (K, n)), 2))  # rounded: plenty of ties
            tracks = [np.empty(n) for _ in range(6)]
            df, win = ctypes.c_double(), ctypes.c_int()
            assert ref(m.ctypes.data_as(dp), K, n, 1.0, 5.0, 0.0, 0, 31, 0.01,
                       *[t.ctypes.data_as(dp) for t in tracks], ctypes.byref(df), ctypes.byref(win)) == 0
            got = oracle.score_centered_wls(m)
            want = (tracks[5], tracks[0], tracks[1], tracks[2], tracks[3], tracks[4])
            for g, w in zip(got[:6], want):
                assert g.tobytes() == w.tobytes(), (n, K)
            assert got[6] == df.value and got[7] == win.value


def test_oracle_score_loci_wls_known_answers(oracle):
    """The reference's own expectations for `score_loci_wls` (tests/test_rocco.py:235-260)."""
    scores, details = oracle.score_loci_wls(np.array([[1.0, 15.0]]), lower_bound_z=0.0)
    assert details["input_scale"] == "log2p1"
    assert np.allclose(details["mean"], np.array([-1.5, 1.5]))
    assert np.allclose(details["z_scores"], np.array([-0.67449076, 0.67449076]))
    assert np.allclose(scores, np.array([-0.67449076, 0.67449076]))
    scores, details = oracle.score_loci_wls(np.array([[1.0, 15.0]]), min_effect=0.5)
    assert np.isclose(details["min_effect"], 0.5)
    assert scores[1] < details["z_scores"][1] and scores[0] < details["z_scores"][0]


def test_oracle_reproduces_the_reference_score_loci_wls(oracle):
    """tests/golden/score_loci_wls_vectors.npz was written by the reference's own `score_loci_wls`."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "score_loci_wls_vectors.npz"))
    tracks = ("mean", "raw_variance", "prior_variance", "moderated_variance", "standard_error", "z_scores",
              "degrees_of_freedom", "centered_matrix")
    for name in gold["names"]:
        lbz, pdf, me, pfr = gold[f"{name}_params"]
        kw = dict(lower_bound_z=lbz, prior_df=pdf, min_effect=None if np.isnan(me) else me, precision_floor_ratio=pfr)
        scores, details = oracle.score_loci_wls(gold[f"{name}_counts"], **kw)
        assert scores.tobytes() == gold[f"{name}_scores"].tobytes(), name
        for key in tracks:
            assert np.asarray(details[key], dtype=np.float64).tobytes() == gold[f"{name}_{key}"].tobytes(), (name, key)
        scalars = np.array([details["local_baseline_window"], details["local_baseline_lambda"], details["min_effect"],
                            details["precision_floor_ratio"], details["prior_spatial_window"]], dtype=np.float64)
        assert np.array_equal(scalars, gold[f"{name}_scalars"]), name


def test_oracle_summit_offsets_known_answer(oracle):
    """The reference's expectation for `_write_narrowpeak_summit_offsets` (tests/test_rocco.py:301-325)."""
    tracks = {"chr1": (np.array([100, 150, 200]), np.array([125, 175, 225]), np.array([1.0, 5.0, 2.0], dtype=np.float32)),
              "chr2": None}
    got = oracle.narrowpeak_summit_offsets([("chr1", 100, 250), ("chr1", 150, 250), ("chr2", 0, 50)], tracks)
    assert got == [("chr1_100_250", 75), ("chr1_150_250", 25), ("chr2_0_50", -1)]
    assert oracle.narrowpeak_summit_track(np.array([5]), np.array([1.0])) is None


def test_oracle_reproduces_the_reference_budget_null_draws(oracle):
    """tests/golden/budget_null_vectors.npz was written by the reference's `_fit_budget_null_residual_template` and
    `_compute_budget_null_draw` (with the multipliers of each draw stored)."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "budget_null_vectors.npz"))
    for name in gold["names"]:
        lbz, pdf, me, pfr, center, soft_scale, threshold = gold[f"{name}_params"]
        me = None if np.isnan(me) else float(me)
        template, scores, positive = oracle.fit_budget_null_residual_template(gold[f"{name}_centered"], lbz, pdf, me, pfr)
        assert template.tobytes() == gold[f"{name}_template"].tobytes(), name
        assert scores.tobytes() == gold[f"{name}_fitted_scores"].tobytes() and positive.tobytes() == gold[f"{name}_positive"].tobytes()
        for draw in range(2):
            stats = oracle.compute_budget_null_draw(template, gold[f"{name}_draw{draw}_weights"], lbz, pdf, me, pfr, center,
                                                    soft_scale, threshold)
            assert np.array(stats).tobytes() == gold[f"{name}_draw{draw}_stats"].tobytes(), (name, draw)
