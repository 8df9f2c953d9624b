cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_exp_cur
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_exp_cur -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/round_probe.py 62e6 > $GRAFT_REPO_ROOT/gpurun_out/prof_exp_cur/out.log 2>&1
