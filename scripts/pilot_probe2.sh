# Ad-hoc: pilot rounds x penalties, alternating, whole genome and the N = 8 shard
for i in 1 2; do
 for cfg in "2 32" "2 48" "2 64" "3 32" "2 40"; do
  set -- $cfg
  export ROCCO_HIP_PILOT_ROUNDS=$1 ROCCO_HIP_PILOT_POINTS=$2
  g=$(timeout -k 10 120 python bench.py --headline-only --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  s=$(timeout -k 10 120 python bench.py --headline-only --chroms chr1,chr15,chr21 --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "pilot $1 x $2: genome $g shard $s"
 done
done
