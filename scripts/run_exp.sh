cd /tmp && export TMPDIR=/tmp
for v in k1exp3; do
  export ROCCO_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_exp/librocco_$v.so
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$v
  rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$v -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/round_probe.py 62e6 > $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$v/out.log 2>&1
done
