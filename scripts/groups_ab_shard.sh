# Ad-hoc: groups = 1 against 3 on the shards rank 0 owns at N = 2 / 4 / 8, alternating
for c in chr1,chr4,chr5,chrX,chr9,chr11,chr13,chr14,chr16,chr18,chrY,chr21 chr1,chrX,chr10,chr15,chr17,chr21 chr1,chr15,chr21; do
 for i in 1 2 3; do
  for g in 1 3; do
   ROCCO_SOLVE_GROUPS=$g timeout -k 10 120 python bench.py --headline-only --chroms $c --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c'.count(',')+1, 'chromosomes groups=$g', d['ms_per_step'])"
  done
 done
done
