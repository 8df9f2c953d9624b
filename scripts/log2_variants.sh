# Ad-hoc: the log-scale kernel with the two-stage log2 (default) and with the full evaluation only, same box
set -e
cd "$GRAFT_REPO_ROOT"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
for v in "" "-DROCCO_LOG2_FULL_ONLY"; do
  touch rocco_amd/csrc/wls.hip
  make -C rocco_amd/csrc CXXFLAGS="$BASE $v" > /dev/null 2>&1
  echo "== variant '$v'"
  timeout -k 10 200 python scripts/log2_probe.py 2>&1 | grep -v amdgpu.ids
done
touch rocco_amd/csrc/wls.hip; make -C rocco_amd/csrc > /dev/null 2>&1
