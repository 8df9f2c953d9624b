# Ad-hoc: fewer penalties per workgroup on rounds that fit the device at once (ROCCO_HIP_LEAN_BATCH=1) against always 8 / 4
for c in chr1,chr15,chr21 chr1,chrX,chr10,chr15,chr17,chr21 all; do
 for i in 1 2 3; do
  for v in 1 0; do
   if [ $c = all ]; then arg=""; else arg="--chroms $c"; fi
   ROCCO_HIP_LEAN_BATCH=$v timeout -k 10 120 python bench.py --headline-only $arg --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c'[:12], 'adaptive=$v', d['ms_per_step'])"
  done
 done
done
