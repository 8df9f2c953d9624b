# Ad-hoc (round 5): kernel durations of the segmented baseline sweeps (whole genome, K = 100) for one build variant.
#   G=32 H=4 bash scripts/whittaker_segment_profile.sh
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05/prof
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
touch rocco_amd/csrc/whittaker.hip
make -C rocco_amd/csrc CXXFLAGS="$BASE -DROCCO_GROUP_ROWS=${G:-8} -DROCCO_ROW_HELPERS=${H:-4}" > /dev/null 2>&1
export ROCCO_HIP_WHITTAKER_TRACE=1
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/r05/prof"; rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r05/prof" -o wseg -- python3 "$GRAFT_REPO_ROOT/scripts/whittaker_batch_probe.py" ${PROBE_ARGS:-100 all} > /tmp/wseg_probe.txt 2>&1 || true
grep -v "amdgpu.ids\|simple_timer" /tmp/wseg_probe.txt | tail -n 12
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r05/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "whittaker" in r["Name"]:
            print(r["Name"][:90], "calls", r["Calls"], "avg ms", float(r["AverageNs"]) / 1e6, "max ms", float(r["MaxNs"]) / 1e6)
PY
touch rocco_amd/csrc/whittaker.hip; make -C rocco_amd/csrc > /dev/null 2>&1
