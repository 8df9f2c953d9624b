# Ad-hoc: calibration time of a small shard and the whole-genome step under different pilot / point settings
mkdir -p gpurun_out/r02
for cfg in "3 16" "2 32" "2 64" "1 64"; do
  set -- $cfg
  export ROCCO_HIP_PILOT_ROUNDS=$1 ROCCO_HIP_PILOT_POINTS=$2
  echo "== pilot rounds $1 points $2"
  timeout -k 10 120 python scripts/calib_timeline.py chr1,chr15,chr21 2>&1 | grep calibrate
  timeout -k 10 120 python scripts/calib_timeline.py chr1 2>&1 | grep calibrate
  timeout -k 10 120 python bench.py --headline-only --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'])"
done
