"""Ad-hoc (round 5): bench.py's solve_by_track_kind leg on its own.   python scripts/track_kind_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import argparse, torch
import bench
import pyoracle as po
from rocco_amd import synth
po.build()
args = argparse.Namespace(seed=20240, step_bp=50)
genome = synth.chrom_loci(50, None)
leg = bench._solve_by_track_kind_leg(args, genome, torch.device("cuda:0"), po)
for name, e in leg.items():
    print(f"{name:38s} zeros {e['exact_zero_fraction']:.3f}  chr1 {e['chr1']['ms']:7.3f} ms passes {e['chr1']['passes_max']:2d} paths {e['chr1']['paths']}  "
          f"genome {e['genome']['ms']:7.3f} ms ({e['genome']['vs_hash_tracks']:.2f}x) passes {e['genome']['passes_max']:2d} paths {e['genome']['paths']} maps {e['genome']['maps']} "
          f"zone {e['genome']['zone_iters']}  oracle {e['oracle_check']}")
print(json.dumps(leg))
