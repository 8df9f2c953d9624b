"""Ad-hoc fuzzing: count-path kernels (rows a2-a4) on the GPU against the CPU oracle, bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle as po
from rocco_amd import inference
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + seconds
t_said = time.time()
it = bad = 0
counts = {}
while time.time() < t_end:
    rng = np.random.default_rng(770000 + it)
    K = int(rng.choice([1, 2, 3, 5, 16, 17, 40]))
    n = int(rng.choice([5, 24, 25, 26, 31, 32, 63, 64, 500, 511, 512, 513, 1025, 4000, 8191, 8192, 8193, 30000, 100000]))
    which = rng.choice(["baseline", "wls", "loci_pow2", "batch", "batch"])
    os.environ["ROCCO_HIP_ROLLING_GROUP"] = str(rng.choice([0, 1, 2, 4, 8]))  # (0: the library's own choice)
    # round 5: rows of the batched baselines cut into segments that start from a warm-up and are verified at their seams --
    # forced short here (the library cuts only rows far longer than these), warm-ups short enough that seams get repaired
    for name, choices in (("ROCCO_HIP_WHITTAKER_SEGMENT_LOCI", [None, 0, 1024, 4096]), ("ROCCO_HIP_WHITTAKER_WARM_LOCI", [None, 64, 512, 4096])):
        pick = choices[int(rng.integers(0, len(choices)))]
        os.environ.pop(name, None)
        if pick is not None:
            os.environ[name] = str(pick)
    try:
        if which == "batch":
            # several matrices through the batched launches (grouped baselines, grouped rolling sums, pipelines) against the
            # oracle's baselines and against the single-matrix scoring
            import torch
            count = int(rng.integers(2, 7))
            mats, lam = [], inference._consenrich_whittaker_lambda(101)
            for _ in range(count):
                Ki = int(rng.choice([1, 2, 7, 8, 9, 16, 17, 33]))
                ni = int(rng.choice([101, 127, 128, 129, 640, 4095, 4096, 4100, 20000, 70001]))
                mats.append(np.ldexp(1.0, rng.integers(0, 9, (Ki, ni))) - 1.0)
            dev = [torch.from_numpy(m).to("cuda:0") for m in mats]
            logs = [torch.log2(d + 1.0) for d in dev]
            bases = inference.crossfit_whittaker_baseline_batch_device(logs, lam)
            ok = all(b.cpu().numpy().tobytes() == po.crossfit_whittaker_baseline(l.cpu().numpy(), lam).tobytes() for b, l in zip(bases, logs))
            batch = inference.score_loci_wls_batch_device(dev, workers=int(rng.integers(1, 4)))
            for d, (sc, det) in zip(dev, batch):
                one, one_det = inference.score_loci_wls_device(d)
                ok = ok and torch.equal(sc, one) and all(torch.equal(det[x], one_det[x]) for x in ("mean", "standard_error", "centered_matrix"))
            o0, _ = po.score_loci_wls(mats[0])
            ok = ok and batch[0][0].cpu().numpy().tobytes() == o0.tobytes()
            K, n = len(mats), max(m.shape[1] for m in mats)
        elif which == "baseline":
            m = rng.normal(0, rng.choice([1e-6, 1.0, 1e4]), (K, n)); m[rng.random(m.shape) < 0.2] = 0.0
            lam = float(rng.choice([0.01, 7.0 * (3 * 0.15915494) ** 4, inference._consenrich_whittaker_lambda(101), 1e9]))
            ok = inference.crossfit_whittaker_baseline(m, lam).tobytes() == po.crossfit_whittaker_baseline(m, lam).tobytes()
        elif which == "wls":
            scale = rng.choice([1e-3, 1.0, 50.0])
            m = rng.normal(0, scale, (K, n)) * np.exp(rng.normal(0, 0.7, (1, n)))
            if rng.random() < 0.4: m = np.round(m, int(rng.integers(0, 3)))
            if rng.random() < 0.2: m[rng.integers(0, K)] = 0.0
            kw = dict(lower_bound_z=float(rng.choice([0.0, 1.0])), prior_df=float(rng.choice([0.0, 5.0, 50.0])),
                      min_effect=(None if rng.random() < 0.5 else float(rng.random())),
                      spatial_window=int(rng.choice([5, 6, 7, 31, 33, 62, 63])),
                      precision_floor_ratio=float(rng.choice([0.0, 0.01, 1.0])))
            g, o = inference.score_centered_wls(m, **kw), po.score_centered_wls(m, **kw)
            ok = all(a.tobytes() == b.tobytes() for a, b in zip(g[:6], o[:6])) and g[6:] == o[6:]
        else:
            k = rng.integers(0, 8, (K, n)); c = np.ldexp(1.0, k) - 1.0
            g, gd = inference.score_loci_wls(c, return_details=True); o, od = po.score_loci_wls(c)
            ok = g.tobytes() == o.tobytes() and all(np.asarray(gd[x]).tobytes() == np.asarray(od[x]).tobytes()
                                                       for x in ("mean", "standard_error", "centered_matrix", "prior_variance"))
    except Exception as exc:
        ok = False; which = f"{which}: {type(exc).__name__} {exc}"
    counts[which] = counts.get(which, 0) + 1
    if not ok:
        bad += 1; print(f"MISMATCH it={it} K={K} n={n} {which}", flush=True)
    it += 1
    if time.time() - t_said > 60.0:  # (a GPU box takes a command that says nothing for minutes to be hung)
        t_said = time.time(); print(f"... {it} cases, {bad} mismatches so far", flush=True)
from rocco_amd import _native
print(f"{it} cases, {bad} mismatches; {sorted(counts.items())}; seams repaired along the way: {int(_native.load().rocco_hip_whittaker_seam_repairs())}")
