# Round 5: LDS bank conflicts of the count path's two chain kernels (SQ_LDS_BANK_CONFLICT = extra cycles, SQ_LDS_IDX_ACTIVE = all
# LDS-array cycles; one PMC pass per program) -- the batched baseline sweeps (residual form) and the batched rolling sums, K = 100 genome
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
rm -rf /tmp/pmc_lds_w /tmp/pmc_lds_r
PROBE_RESIDUAL=1 timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_lds_w -- python3 scripts/whittaker_batch_probe.py 100 all > $O/pmc_lds_w.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_lds_r -- python3 scripts/rolling_rows_probe.py > $O/pmc_lds_r.log 2>&1
python3 - <<'PY' | tee $O/lds_conflicts.txt
import csv, glob, re
print("LDS bank conflicts (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE), K = 100 genome, summed over the dispatches of each kernel:")
for d in ("/tmp/pmc_lds_w", "/tmp/pmc_lds_r"):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "whittaker_rows_kernel" not in n and "wls_rolling_rows_kernel" not in n and "whittaker_seam" not in n:
            continue
        short = re.sub(r"\(.*", "", n.replace("rocco::(anonymous namespace)::", "").replace("void ", ""))
        tot.setdefault(short, {}).setdefault(r["Counter_Name"], 0.0)
        tot[short][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(tot.items()):
        c, a = v.get("SQ_LDS_BANK_CONFLICT", 0.0), v.get("SQ_LDS_IDX_ACTIVE", 0.0)
        print(f"  {k:45s} conflict cycles {c:16.0f}   LDS-array cycles {a:16.0f}   conflicts / active = {c / max(a, 1.0):.4f}")
PY
