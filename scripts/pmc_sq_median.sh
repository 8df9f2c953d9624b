# SQ / GRBM counters of the whole-genome median launch (explain the un-overlapped issue time).
# Separate passes per counter group (8 SQ slots per pass); never combined with trace domains other than the kernel trace.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r03}
mkdir -p gpurun_out/$tag
rocprofv3 -L > gpurun_out/$tag/counters_available.txt 2>&1 || true
pass() {
  name=$1; shift
  rm -rf gpurun_out/$tag/pmc_$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_$name -- python3 scripts/pmc_median.py > gpurun_out/$tag/pmc_$name.log 2>&1
  f=$(ls gpurun_out/$tag/pmc_$name/*/*counter_collection.csv | head -1)
  python3 - "$f" "$name" "$tag" <<'PY'
import csv, sys, collections
path, name, tag = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "median_batch_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"gpurun_out/{tag}/sq_{name}.txt", "w") as fh:
    for k, v in sorted(acc.items()):
        line = f"{k}: mean {sum(v)/len(v):.6g} over {len(v)} launches"
        print(line); fh.write(line + "\n")
PY
  rm -rf gpurun_out/$tag/pmc_$name
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
pass c GRBM_GUI_ACTIVE GRBM_COUNT
