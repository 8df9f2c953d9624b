cd /tmp && export TMPDIR=/tmp
for v in base exp1 exp2; do
  if [ $v = base ]; then unset ROCCO_HIP_LIBRARY; else export ROCCO_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_exp/librocco_$v.so; fi
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$v
  rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$v -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/active_probe.py chr14 > $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$v/out.log 2>&1
done
