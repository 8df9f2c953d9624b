# Ad-hoc (round 5): the segmented baseline sweeps with other group sizes / helper counts / segment plans -- whole-genome
# K = 100 probe per variant.   VARIANTS="G:H ..." (rows per workgroup : helper wavefronts)  ENVS="name=value,... ..."
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
VARIANTS=${VARIANTS:-8:4 16:4 32:4 32:8}
ENVS=${ENVS:-X=0}
for v in $VARIANTS; do
  g=$(echo "$v" | cut -d: -f1); h=$(echo "$v" | cut -d: -f2); d=$(echo "$v" | cut -d: -f3 | tr '+' ' ')  # (third field: -D flags, '+' between them)
  touch rocco_amd/csrc/whittaker.hip
  make -C rocco_amd/csrc CXXFLAGS="$BASE -DROCCO_GROUP_ROWS=$g -DROCCO_ROW_HELPERS=$h $d" > /dev/null 2>&1
  for e in $ENVS; do
    echo "== rows per workgroup $g, helpers $h, $d $e"
    env $(echo "$e" | tr ',' ' ') timeout -k 10 200 python scripts/whittaker_batch_probe.py ${PROBE_ARGS:-100 all} 2>&1 | grep -v amdgpu.ids | tail -n 3
  done
done
touch rocco_amd/csrc/whittaker.hip; make -C rocco_amd/csrc > /dev/null 2>&1
