"""Ad-hoc (round 5): the composed driver's count branch (device multipliers) with the null draws one at a time / several at
once, inputs kept / handed over: seconds and whether cache entries and BED bytes agree.  python scripts/null_batch_probe.py [chroms] [K]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, rocco as rr
names = (sys.argv[1] if len(sys.argv) > 1 else "chr19,chr20,chr21,chr22").split(",")
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
genome = synth.chrom_loci(50, None)
index = {n: i for i, (n, _) in enumerate(genome)}
dev = torch.device("cuda:0")
base = {"budget_null_draws": 25, "threads": -1, "gamma": None, "budget": None, "scale_chrom_budgets": 1.0, "budget_posterior_quantile": 0.01,
        "selection_penalty": None, "min_length_bp": None, "score_lower_bound_z": 1.0, "score_prior_df": 5.0, "score_min_effect": None,
        "score_precision_floor_ratio": 0.01, "low_memory": False, "narrowPeak": False, "input_track_type": "bam", "budget_null_multipliers": "device"}
def inputs():
    out = {}
    for name in names:
        i = index[name]; n = genome[i][1]
        m = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, i), device=dev); m.mul_(20.0).round_()
        out[name] = (np.arange(n, dtype=np.int64) * 50, m)
    return out
tmp = tempfile.mkdtemp(); os.chdir(tmp)
ref = None
for label, env, consume in (("one draw at a time", "1", False), ("draws together", "", False), ("draws together, inputs handed over", "", True), ("one at a time again", "1", False)):
    os.environ["ROCCO_BUDGET_NULL_DRAWS_AT_ONCE"] = env
    ins = inputs()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cache = rr._build_chrom_cache(names, ins, dict(base, consume_inputs=consume))
    torch.cuda.synchronize(); t_cache = time.perf_counter() - t0
    key = {c: (cache[c]["budget_count_hat"], cache[c]["budget_fraction_hat"], cache[c]["gamma"], cache[c]["budget_rate_meta"].get("num_null_draws")) for c in names}
    if ref is None:
        ref = key
    print(f"{label:40s}: cache {t_cache:.3f} s, same estimates as the first run: {key == ref}", flush=True)
    if key != ref:
        for c in names:
            if key[c] != ref[c]:
                print("   ", c, key[c], "vs", ref[c])
    del ins, cache
