# Ad-hoc (round 5): per-kernel totals of the composed driver, K = 100 count matrices, whole genome, device multipliers
set -e
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out/r05"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cd
PROBE_REPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cd -o cd -- python3 "$R/scripts/composed_probe.py" counts 100 all device > "$R/gpurun_out/r05/composed_prof.txt" 2>&1 || true
grep "^{" "$R/gpurun_out/r05/composed_prof.txt" || true
cp /tmp/prof_cd/cd_kernel_stats.csv "$R/gpurun_out/r05/composed_kernel_stats.csv"
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("/tmp/prof_cd/cd_kernel_stats.csv")))
print("total kernel ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6)
for r in rows[:30]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.1f} ms {int(r['Calls']):6d} {r['Name'][:100]}")
PY
