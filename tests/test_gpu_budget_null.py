"""GPU: the K x n part of the wild-bootstrap budget null (rocco/inference.py:628-722)
through the C ABI: NumPy-order sums against NumPy itself, the residual template and the per-draw statistics against
golden vectors written by the reference's own functions (tests/golden/make_golden_budget_null.py): bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "budget_null_vectors.npz")


@pytest.mark.parametrize("n", [1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16384, 20000, 100003, 1 << 20])
def test_numpy_sum_order(gpu, n):
    import torch

    from rocco_amd.inference import numpy_sum_device

    rng = np.random.default_rng(n)
    for x in (rng.normal(size=n), rng.gamma(0.5, 3.0, size=n) * 10.0 ** rng.integers(-8, 8, size=n), np.full(n, 0.1)):
        got = numpy_sum_device(torch.from_numpy(x).cuda())
        assert np.float64(got).tobytes() == np.float64(np.sum(x)).tobytes(), n
        assert np.float64(got / n).tobytes() == np.float64(np.mean(x)).tobytes(), n


def test_golden_residual_template_and_draws(gpu):
    import torch

    from rocco_amd.inference import compute_budget_null_draw_device, fit_budget_null_residual_template_device

    gold = np.load(GOLD)
    for name in gold["names"]:
        name = str(name)
        lbz, pdf, me, pfr, center, soft_scale, threshold = gold[f"{name}_params"]
        me = None if np.isnan(me) else float(me)
        centered_t = torch.from_numpy(gold[f"{name}_centered"]).cuda()
        template_t, scores_t, positive_t = fit_budget_null_residual_template_device(centered_t, lbz, pdf, me, pfr)
        assert template_t.cpu().numpy().tobytes() == gold[f"{name}_template"].tobytes(), name
        assert scores_t.cpu().numpy().tobytes() == gold[f"{name}_fitted_scores"].tobytes(), name
        assert positive_t.cpu().numpy().tobytes() == gold[f"{name}_positive"].tobytes(), name
        for draw in range(2):
            weights_t = torch.from_numpy(gold[f"{name}_draw{draw}_weights"]).cuda()
            stats = compute_budget_null_draw_device(template_t, weights_t, lbz, pdf, me, pfr, center, soft_scale, threshold)
            assert np.array(stats).tobytes() == gold[f"{name}_draw{draw}_stats"].tobytes(), (name, draw)


def test_large_draw_matches_the_oracle(gpu, oracle):
    import torch

    from rocco_amd.inference import compute_budget_null_draw_device, fit_budget_null_residual_template_device

    rng = np.random.default_rng(8)
    K, n = 5, 150000
    centered = rng.normal(0.0, 0.6, size=(K, n)) + 2.0 * (rng.random((1, n)) < 0.03)
    weights = rng.normal(size=(K, n))
    o_template, o_scores, _ = oracle.fit_budget_null_residual_template(centered)
    template_t, scores_t, _ = fit_budget_null_residual_template_device(torch.from_numpy(centered).cuda())
    assert template_t.cpu().numpy().tobytes() == o_template.tobytes() and scores_t.cpu().numpy().tobytes() == o_scores.tobytes()
    center, soft = float(np.median(o_scores)), 0.8
    want = oracle.compute_budget_null_draw(o_template, weights, 1.0, 5.0, None, 0.01, center, soft, center + 1.2)
    got = compute_budget_null_draw_device(template_t, torch.from_numpy(weights).cuda(), 1.0, 5.0, None, 0.01, center, soft,
                                          center + 1.2)
    assert np.array(got).tobytes() == np.array(want).tobytes()
