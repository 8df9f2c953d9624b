"""Ad-hoc: phase timers of K3 (library built with the g_prof instrumentation, ROCCO_HIP_LIBRARY)."""
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
wg, sl = max(buf[3], 1), max(buf[7], 1)
print(f"K3 workgroups {wg}, slot passes {sl} ({sl/wg:.2f} per workgroup): staging {buf[1]*10/wg:.0f} ns per workgroup; per slot pass: "
      f"before the recursion (incoming delta / clear index scans, mode) {buf[4]*10/sl:.0f} ns, recursion {buf[5]*10/sl:.0f} ns, "
      f"after it (reductions, stores, fill summaries) {buf[6]*10/sl:.0f} ns; of the first: issuing the loads {buf[8]*10/sl:.0f} ns, "
      f"incoming delta (waits for them, scan, two barriers) {buf[9]*10/sl:.0f} ns, mode + clear-index scan {buf[10]*10/sl:.0f} ns")
