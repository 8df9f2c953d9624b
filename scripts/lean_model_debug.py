"""Ad-hoc: how many penalties the lean model kernel leaves open, by array length / kind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import pyoracle as po
from rocco_amd import delta
rng = np.random.default_rng(1)
for n in [8191, 8192, 8193, 8224, 8225, 16384, 16385, 30000, 100000]:
    s = np.round(rng.gamma(1.0, 0.3, n), 5)
    s[rng.integers(0, n, max(1, n // 40))] += np.round(rng.gamma(6.0, 1.0, max(1, n // 40)), 5)
    s_t = torch.from_numpy(s).cuda()
    lam_ref = float(np.quantile(s, 0.9))
    margin = 1.0 + float(np.ptp(s)) + float(np.max(np.abs(s))) + 4.0
    emap = delta.delta_build_map_device(s_t, 1.0, lam_ref, margin)
    codes = emap.cpu().numpy()
    lams = list(lam_ref + 1e-3 * rng.uniform(-1, 1, 16))
    got = delta.delta_model_lean_device(s_t, 1.0, lams, emap)
    costs = po.build_switch_costs(s, 1.0)
    wrong = sum(1 for lam, (c, o) in zip(lams, got) if c != po.solve_penalized_chain(s, costs, lam)[2])
    print(n, "open", sum(1 for _c, o in got if o), "reasons", sorted(set(o for _c, o in got)), "of", len(lams), "wrong counts", wrong, "hazard chunks", int((codes & 0x80 != 0).sum()), "of", codes.size,
          "last codes", codes[-3:].tolist())
