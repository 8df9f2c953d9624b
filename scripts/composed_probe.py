"""Ad-hoc: the composed driver (rocco_amd.rocco.run_chromosomes) on synthetic genome-sized inputs, seconds by phase.
    python scripts/composed_probe.py bigwig 10 all device|host [draws]
    python scripts/composed_probe.py counts 100 chr20,chr21,chr22 device"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, budget
from rocco_amd import rocco as rr

branch, K, which, mult = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
draws = int(sys.argv[5]) if len(sys.argv) > 5 else 25
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
names = [n for n, _ in genome] if which == "all" else which.split(",")
index = {name: i for i, (name, _n) in enumerate(genome)}
consume = bool(os.environ.get("PROBE_CONSUME"))  # (args["consume_inputs"]: the matrices are made afresh for every repetition)


def make_inputs():
    made = {}
    for name in names:
        n = genome[index[name]][1]
        m = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, index[name]), device=device)
        if branch == "counts":
            m.mul_(20.0).round_()
        made[name] = (np.arange(n, dtype=np.int64) * 50, m)
    torch.cuda.synchronize()
    return made


inputs = make_inputs()
args = {"input_track_type": "bigwig" if branch == "bigwig" else "bam", "budget_null_draws": draws, "threads": -1, "gamma": None,
        "budget": None, "scale_chrom_budgets": 1.0, "budget_posterior_quantile": 0.01, "selection_penalty": None,
        "min_length_bp": None, "score_lower_bound_z": 1.0, "score_prior_df": 5.0, "score_min_effect": None,
        "score_precision_floor_ratio": 0.01, "low_memory": False, "narrowPeak": False, "budget_null_multipliers": mult}
home = os.getcwd()
with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
        if consume:
            args["consume_inputs"] = True
            if rep > 0:
                inputs = None
                inputs = make_inputs()
        phases, mults = {}, {}
        budget.collect_timings(mults)
        args["_phase_seconds"] = phases
        args["output"] = os.path.join(tmp, f"out{rep}.bed")
        t0 = time.perf_counter()
        out = rr.run_chromosomes(names, inputs, dict(args), run_id=str(rep))
        total = time.perf_counter() - t0
        budget.collect_timings(None)
        lines = sum(1 for _ in open(out))
        print(json.dumps({"branch": branch, "K": K, "chromosomes": len(names), "loci": sum(genome[index[n]][1] for n in names),
                          "multipliers": mult, "draws_max": draws, "seconds": round(total, 3), "intervals": lines,
                          "phases": {k: round(v, 3) for k, v in phases.items()},
                          "multipliers_seconds": {k: round(v, 3) for k, v in mults.items()},
                          "consume_inputs": consume, "max_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}), flush=True)
    os.chdir(home)  # (a profiler that finalises in a deleted directory aborts)
