# Ad-hoc (round 5): where the rolling-sums kernel's cycles go, by compiling parts out (results are garbage: timing only)
#   VARIANTS="flags;flags;..."  (';' between variants; ROCCO_ROLL_FLUSH=8: the chain wavefront stores one sum per batch of 8,
#   ROCCO_ROLL_NOVAR: no variances, ROCCO_ROLL_NOCHAIN: helpers only)
set -e
cd "$GRAFT_REPO_ROOT"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
IFS=';' read -ra LIST <<< "${VARIANTS:--DROCCO_ROLL_FLUSH=1;-DROCCO_ROLL_FLUSH=8;-DROCCO_ROLL_FLUSH=8 -DROCCO_ROLL_NOVAR;-DROCCO_ROLL_NOCHAIN}"
for v in "${LIST[@]}"; do
  touch rocco_amd/csrc/wls.hip
  make -C rocco_amd/csrc CXXFLAGS="$BASE -DROCCO_ROLL_STAMPS $v" > /dev/null 2>&1
  echo "== $v"
  timeout -k 10 200 python scripts/rolling_rows_probe.py 2>&1 | grep -v amdgpu | tail -n 4
done
touch rocco_amd/csrc/wls.hip; make -C rocco_amd/csrc > /dev/null 2>&1
