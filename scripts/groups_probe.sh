# Ad-hoc: whole-genome step under different group counts / score-first settings (same box, back to back)
for cfg in "3 0" "1 0" "2 0" "4 0" "1 1" "2 1" "3 1" "3 0"; do
  set -- $cfg
  ROCCO_SOLVE_GROUPS=$1 ROCCO_SCORE_FIRST=$2 timeout -k 10 120 python bench.py --headline-only --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('groups=$1 score_first=$2', d['ms_per_step'])"
done
