# Ad-hoc (round 5): per-kernel totals of the whole-genome K = 100 count path (score_loci_wls_batch_device) with W pipelines.
#   W=1 bash scripts/count_batch_kernel_stats.sh
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
export PROBE_BATCH_ONLY=1 PROBE_REPS=${REPS:-3} ROCCO_BATCH_TRACE=1 ROCCO_HIP_WHITTAKER_TRACE=1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cb -o cb -- python3 "$GRAFT_REPO_ROOT/scripts/count_genome_batch_probe.py" 100 all ${W:-1} > /tmp/cb_probe.txt 2>&1 || true
grep "^batch\|^\[batch\]\|^\[whittaker\]" /tmp/cb_probe.txt | tail -n 14
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import csv, glob, os
for f in glob.glob("/tmp/prof_cb/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    reps = int(os.environ.get("PROBE_REPS", "3"))
    print(f"kernel totals per call (ms, {reps} calls in the file; + the probe's own set-up kernels):")
    for r in rows[:22]:
        print(f"  {float(r['TotalDurationNs']) / 1e6 / reps:9.2f}  calls {int(r['Calls']) // reps:6d}  {r['Name'][:110]}")
    import shutil
    shutil.copy(f, f"gpurun_out/r05/count_batch_kernel_stats_w{os.environ.get('W', '1')}.csv")
PY
