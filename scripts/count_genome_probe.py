"""Ad-hoc: count-path pipeline (score_loci_wls -> solve -> decode) over several chromosomes, groups side by side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
names = (sys.argv[2] if len(sys.argv) > 2 else "chr15,chr16,chr17,chr18,chr19,chr20,chr21,chr22").split(",")
genome = synth.chrom_loci(50, names)
works = [pipeline.ChromWork(name, (synth.hash_matrix_device(K, n, 100 + i) * 20.0).contiguous(), 0.02, 1.0, step=50, scoring="wls")
         for i, (name, n) in enumerate(genome)]
loci = sum(n for _, n in genome)
ref = None
for g in (1, 2, 4, 8, 1, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = pipeline.solve_rank(works, groups=g)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    sig = [(r["selection_penalty"], r["selected_count"]) for r in res]
    if ref is None: ref = sig
    assert sig == ref
    print(f"K={K} {len(works)} chromosomes {loci} loci, groups={g}: {t*1e3:.1f} ms ({loci/t/1e6:.2f} M loci/s)", flush=True)
