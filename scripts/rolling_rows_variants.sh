# Ad-hoc: the rolling rows kernel with either half compiled out (timing only; results are wrong in those builds)
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
for v in "" "-DROCCO_ROLL_NOVAR" "-DROCCO_ROLL_NOCHAIN"; do
  touch rocco_amd/csrc/wls.hip
  make -C rocco_amd/csrc CXXFLAGS="$BASE $v" > /dev/null 2>&1
  echo "== variant '$v'"
  timeout -k 10 200 python scripts/rolling_rows_probe.py
done
touch rocco_amd/csrc/wls.hip; make -C rocco_amd/csrc > /dev/null 2>&1
