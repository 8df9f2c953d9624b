# Round-3 profiles (run on the GPU box; results under gpurun_out/r03/, copied into profiles/ by hand afterwards):
#   bench under rocprofv3 (kernel stats of the headline), the calibration's kernel timeline, the count-path batch's kernel stats
set -e
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03"; mkdir -p "$O"
rm -rf /tmp/prof_bench /tmp/prof_calib /tmp/prof_count
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python3 "$R/bench.py" --steps 20 --warmup 5 --headline-only > "$O/bench_line_under_rocprof.json" 2> "$O/bench_under_rocprof.err"
cp /tmp/prof_bench/b_kernel_stats.csv "$O/kernel_stats.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_calib -o c -- python3 "$R/scripts/step_phases.py" > "$O/calib_tl.log" 2>&1
python3 "$R/scripts/last_calibration.py" /tmp/prof_calib/c_kernel_trace.csv > "$O/calibration_timeline.txt"
PROBE_BATCH_ONLY=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_count -o cb -- python3 "$R/scripts/count_genome_batch_probe.py" 100 all 3 > "$O/count_batch_prof.log" 2>&1
cp /tmp/prof_count/cb_kernel_stats.csv "$O/count_path_batch_kernel_stats.csv"
tail -n 2 "$O/calibration_timeline.txt"; grep "^batch" "$O/count_batch_prof.log"; head -c 600 "$O/bench_line_under_rocprof.json"
