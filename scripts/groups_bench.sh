for cfg in "ROCCO_SOLVE_GROUPS=1" "ROCCO_SOLVE_GROUPS=2 ROCCO_HIP_CHAIN=1" "ROCCO_SOLVE_GROUPS=2 ROCCO_HIP_CHAIN=0" "ROCCO_SOLVE_GROUPS=3 ROCCO_HIP_CHAIN=1" "ROCCO_SOLVE_GROUPS=3 ROCCO_HIP_CHAIN=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python bench.py --steps 30 --warmup 8 --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline'].get('avg_kernel_ms'))"
done
