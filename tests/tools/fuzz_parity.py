"""Ad-hoc fuzzing: random budgeted / fixed-penalty solves on the GPU against the CPU oracle, bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import pyoracle as po
from rocco_amd import dp
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + seconds
t_said = time.time()
it = bad = 0
kinds = {}
while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 100000 + it)
    n = int(rng.choice([1, 2, 3, 5, 31, 32, 33, 100, 1000, 8191, 8192, 8193, 20000, 70000, 300000, 1200000], p=None))
    n = max(1, n + int(rng.integers(-3, 4))) if n > 10 else n
    kind = rng.choice(["normal", "int", "round5", "heavy", "const", "sparse", "tiny", "huge", "offset"])
    if kind == "normal": s = rng.normal(0.2, 1.0, n)
    elif kind == "int": s = rng.integers(-3, 6, n).astype(float)
    elif kind == "round5": s = np.round(rng.gamma(1.0, 0.3, n), 5)
    elif kind == "heavy": s = rng.standard_cauchy(n)
    elif kind == "const": s = np.full(n, float(rng.normal()))
    elif kind == "sparse": s = np.where(rng.random(n) < 0.02, rng.gamma(6.0, 1.0, n), 0.0)
    elif kind == "tiny": s = rng.normal(0, 1e-9, n)
    elif kind == "offset": s = float(rng.choice([1e3, 3e4, -1e6, 1e9])) + rng.gamma(1.0, 1.0, n)
    else: s = rng.normal(0, 1e6, n)
    gamma = float(rng.choice([0.0, 0.5, 1.0, 3.0, 10.0, float(abs(rng.normal()) * 2)]))
    use_vec = n > 1 and rng.random() < 0.25
    costs = rng.gamma(1.0, gamma + 0.1, n - 1) if use_vec else gamma
    o_costs = costs if use_vec else po.build_switch_costs(s, gamma)
    try:
        if rng.random() < 0.5:
            budget = float(rng.choice([0.005, 0.02, 0.05, 0.1, 0.3]))
            target = int(np.floor(n * budget))
            g = dp.calibrate_selection_penalty(s, costs, target)
            o = po.calibrate_selection_penalty(s, o_costs, target)
            # (the value is a difference of sums of magnitude count * (|s| + |penalty|): both sides round there)
            tol = 1e-9 * max(1.0, abs(o[2])) + 8.0 * 2.0 ** -52 * max(1, o[3]) * (float(np.max(np.abs(s))) + abs(o[0]))
            ok = g[0] == o[0] and np.array_equal(g[1], o[1]) and g[3] == o[3] and abs(g[2] - o[2]) <= tol
            mode = "budget"
        else:
            lam = float(rng.choice([0.0, float(np.median(s)), float(rng.normal()), float(np.max(s)) + 1.0, float(np.min(s)) - 1.0]))
            g = dp.solve_penalized_chain(s, costs, lam)
            o = po.solve_penalized_chain(s, o_costs, lam)
            tol = 1e-9 * max(1.0, abs(o[1])) + 8.0 * 2.0 ** -52 * max(1, o[2]) * (float(np.max(np.abs(s))) + abs(lam))
            ok = np.array_equal(g[0], o[0]) and g[2] == o[2] and abs(g[1] - o[1]) <= tol
            mode = "fixed"
    except Exception as exc:  # an error on one side only is a mismatch
        ok, mode = False, f"exception {type(exc).__name__}: {exc}"
    kinds[(mode, kind)] = kinds.get((mode, kind), 0) + 1
    if not ok:
        bad += 1
        print(f"MISMATCH it={it} n={n} kind={kind} gamma={gamma} vec={use_vec} mode={mode}", flush=True)
    it += 1
    if time.time() - t_said > 60.0:  # (a GPU box takes a command that says nothing for minutes to be hung)
        t_said = time.time(); print(f"... {it} iterations so far", flush=True)
print(f"{it} cases, {bad} mismatches; by (mode, kind): {sorted(kinds.items())}")
