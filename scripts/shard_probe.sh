mkdir -p gpurun_out/r02
for c in chr1,chr15,chr21 chr1,chrX,chr10,chr15,chr17,chr21; do
 for cfg in "3 0" "1 0" "2 0" "1 1" "3 1" "2 1"; do
  set -- $cfg
  ROCCO_SOLVE_GROUPS=$1 ROCCO_SCORE_FIRST=$2 timeout -k 10 120 python bench.py --headline-only --chroms $c --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c'.count(',')+1, 'groups=$1 score_first=$2', d['ms_per_step'])"
 done
done
