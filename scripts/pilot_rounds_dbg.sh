for cfg in "X=1" "ROCCO_HIP_CHAIN_PILOT_TILES=8 ROCCO_HIP_CHAIN_PILOT_WGS=256"; do
  echo "== $cfg"
  env $cfg PROBE_DEBUG=1 timeout -k 10 200 python scripts/chain_probe.py all 2 2>&1 | grep "^\[chain\] problem" | tail -24 | awk '{print $6, $8, $9, $10, $11}' | sort | uniq -c
done
