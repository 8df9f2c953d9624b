# Ad-hoc: groups = 1 against groups = 3 on the whole genome, alternating on one box
for i in 1 2 3 4 5; do
 for g in 1 3; do
  ROCCO_SOLVE_GROUPS=$g timeout -k 10 120 python bench.py --headline-only --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('groups=$g', d['ms_per_step'], d['roofline']['frac'])"
 done
done
