"""Ad-hoc: phase timers of K1 (library built with the g_prof instrumentation, ROCCO_HIP_LIBRARY)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline, _native
genome = synth.chrom_loci(50, None)
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx)), 0.02, 1.0, step=50)
         for idx, (name, n) in enumerate(genome)]
pipeline.solve_rank(works, groups=1)
lib = _native.load()
buf = (ctypes.c_ulonglong * 16)()
lib.rocco_hip_debug_prof(buf, 1)
pipeline.solve_rank(works, groups=1); torch.cuda.synchronize()
lib.rocco_hip_debug_prof(buf, 0)
wg = buf[3]
print(f"K1 workgroups {wg}: per workgroup (thread 0 wall clock, 10 ns ticks): descriptors {buf[0]*10/wg:.0f} ns, staging {buf[1]*10/wg:.0f} ns, slots {buf[2]*10/wg:.0f} ns (of which the chunk loops up to the stores {buf[4]*10/wg:.0f} ns)")
