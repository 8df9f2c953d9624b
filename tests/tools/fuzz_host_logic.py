"""Ad-hoc fuzzing of the product's host-side search (rocco_amd/csrc/search.cpp) on the CPU: the harness of
tests/host_logic runs it over an exact-arithmetic evaluator, with the compaction / tile / pilot-noise / slack
behaviours of the device side switched on through the environment, against the CPU oracle, bit for bit.

    python tests/tools/fuzz_host_logic.py SECONDS SEED      # environment: ROCCO_HOSTLOGIC_{TILE,PILOT,SLACK,NOW}
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import hostlogic

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + seconds
it = bad = 0
paths = {}
while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 100000 + it)
    n = int(rng.choice([5, 33, 100, 1000, 8193, 20000, 70000, 300000]))
    kind = rng.choice(["normal", "int", "round5", "heavy", "sparse", "tiny", "huge", "offset", "gamma"])
    if kind == "normal": s = rng.normal(0.2, 1.0, n)
    elif kind == "int": s = rng.integers(-3, 6, n).astype(float)
    elif kind == "round5": s = np.round(rng.gamma(1.0, 0.3, n), 5)
    elif kind == "heavy": s = rng.standard_cauchy(n)
    elif kind == "sparse": s = np.where(rng.random(n) < 0.02, rng.gamma(6.0, 1.0, n), 0.0)
    elif kind == "tiny": s = rng.normal(0, 1e-9, n)
    elif kind == "gamma": s = rng.gamma(2.0, 0.7, n)
    elif kind == "offset": s = float(rng.choice([1e3, 3e4, -1e6, 1e9])) + rng.gamma(1.0, 1.0, n)
    else: s = rng.normal(0, 1e6, n)
    gamma = float(rng.choice([0.0, 0.5, 1.0, 3.0, 10.0, float(abs(rng.normal()) * 2)]))
    budget = float(rng.choice([0.005, 0.02, 0.05, 0.1, 0.3]))
    target = int(np.floor(n * budget))
    depth = int(rng.choice([1, 2, 3]))
    g = hostlogic.calibrate(s, gamma, target, spec_depth=depth)
    o = po.calibrate_selection_penalty(s, po.build_switch_costs(s, gamma), target)
    ok = g[0] == o[0] and np.array_equal(g[1], o[1]) and g[3] == o[3]
    key = (g[4]["path"], g[4]["compact_n"] >= 0)
    paths[key] = paths.get(key, 0) + 1
    if not ok:
        bad += 1
        print(f"MISMATCH it={it} n={n} kind={kind} gamma={gamma} budget={budget} depth={depth}", flush=True)
    it += 1
print(f"{it} cases, {bad} mismatches; by (path, compacted): {sorted(paths.items())}")
