# Ad-hoc: the 64-chain baseline kernel with 2 / 4 / 8 / 15 helper wavefronts
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
for v in 2 1; do
  touch rocco_amd/csrc/whittaker.hip
  make -C rocco_amd/csrc CXXFLAGS="$BASE -DROCCO_ROW_HELPERS=$v" > /dev/null 2>&1
  echo "== helper wavefronts: $v"
  timeout -k 10 300 python scripts/whittaker_batch_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 4
done
touch rocco_amd/csrc/whittaker.hip; make -C rocco_amd/csrc > /dev/null 2>&1
