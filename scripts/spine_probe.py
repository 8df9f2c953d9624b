import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, dp, rocco as rr
from rocco_amd.delta import delta_build_map_device, delta_spine_device, delta_probe_device
for n in (100000, 1000000):
    m = synth.hash_matrix_device(10, n, seed=1)
    s = rr.score_central_tendency_chrom_device(m)
    lam = 0.27
    t=time.perf_counter(); emap = delta_build_map_device(s, 1.0, lam, 16.0); torch.cuda.synchronize(); print(n,'map',time.perf_counter()-t, 'hazard frac', float((emap>=128).float().mean()))
    for L in (1, 15):
        lams=[lam+1e-12*i for i in range(L)]
        t=time.perf_counter(); c,_ = delta_spine_device(s, 1.0, lams, emap, 0); torch.cuda.synchronize(); print(n,'spine L',L,time.perf_counter()-t, c[:2])
        t=time.perf_counter(); c2 = delta_probe_device(s, 1.0, lams, emap); torch.cuda.synchronize(); print(n,'probe L',L,time.perf_counter()-t, c2[0])
