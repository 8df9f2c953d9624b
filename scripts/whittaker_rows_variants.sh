# Ad-hoc: the 64-chain baseline kernel with either half compiled out (timing only; results are wrong in those builds),
# or with another number of helper wavefronts (HELPERS="2 4 8")
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
for v in ${VARIANTS:-"-DROCCO_ROWS_NOCHAIN" "-DROCCO_ROWS_NOHELP"}; do
  touch rocco_amd/csrc/whittaker.hip
  make -C rocco_amd/csrc CXXFLAGS="$BASE $v" > /dev/null 2>&1
  echo "== variant $v"
  timeout -k 10 300 python scripts/whittaker_batch_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 3
done
touch rocco_amd/csrc/whittaker.hip; make -C rocco_amd/csrc > /dev/null 2>&1
