# Ad-hoc: whole-genome step for different pilot sample sizes (tiles per chromosome), alternating
for t in 16 8 16 8 12 24; do
  ROCCO_HIP_PILOT_TILES=$t timeout -k 10 120 python bench.py --headline-only --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pilot tiles $t step', d['ms_per_step'], [v['passes'] for v in d.get('solve_paths',{}).values()][:6])"
done
