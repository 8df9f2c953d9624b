"""Ad-hoc: the composed driver's count branch with the budget estimates one after another (ROCCO_BUDGET_NULL_STREAMS=1)
and side by side (3): the combined BEDs and every chromosome's budget must be identical; seconds of both."""
import hashlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth
from rocco_amd import rocco as rr

names = (sys.argv[1] if len(sys.argv) > 1 else "chr19,chr20,chr21,chr22,chrY").split(",")
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
index = {name: i for i, (name, _n) in enumerate(genome)}
inputs = {}
for name in names:
    n = genome[index[name]][1]
    m = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, index[name]), device=device)
    m.mul_(20.0).round_()
    inputs[name] = (np.arange(n, dtype=np.int64) * 50, m)
args = {"input_track_type": "bam", "budget_null_draws": 25, "threads": -1, "gamma": None, "budget": None, "scale_chrom_budgets": 1.0,
        "budget_posterior_quantile": 0.01, "selection_penalty": None, "min_length_bp": None, "score_lower_bound_z": 1.0,
        "score_prior_df": 5.0, "score_min_effect": None, "score_precision_floor_ratio": 0.01, "low_memory": False,
        "narrowPeak": False, "budget_null_multipliers": "device"}
digests = {}
with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    for streams in ("1", "3", "1", "3"):
        os.environ["ROCCO_BUDGET_NULL_STREAMS"] = streams
        a = dict(args)
        a["output"] = os.path.join(tmp, f"out{streams}.bed")
        t0 = time.perf_counter()
        out = rr.run_chromosomes(names, inputs, a, run_id=streams)
        seconds = time.perf_counter() - t0
        digest = hashlib.sha256(open(out, "rb").read()).hexdigest()
        print(f"streams {streams}: {seconds:.2f} s, {sum(1 for _ in open(out))} intervals, sha256 {digest[:16]}", flush=True)
        digests.setdefault(streams, digest)
assert digests["1"] == digests["3"], "BEDs differ"
print("identical")
