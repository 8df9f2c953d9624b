# Ad-hoc (round 5): where the wavefronts of the rolling-sums kernel spend their cycles (-DROCCO_ROLL_STAMPS build; workgroup 0)
set -e
cd "$GRAFT_REPO_ROOT"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
touch rocco_amd/csrc/wls.hip
make -C rocco_amd/csrc CXXFLAGS="$BASE -DROCCO_ROLL_STAMPS $EXTRA" > /dev/null 2>&1
timeout -k 10 200 python scripts/rolling_rows_probe.py 2>&1 | grep -v amdgpu | tail -n 8
touch rocco_amd/csrc/wls.hip; make -C rocco_amd/csrc > /dev/null 2>&1
