# Round-5 profiles (run on the GPU box; results under gpurun_out/r05/, copied into profiles/ afterwards):
#   the default bench line; the bench under rocprofv3 (kernel stats of the headline); the chained calibration's kernel
#   timeline; the count-path batch's kernel stats (one and two pipelines); the device normal generator's kernel stats
set -e
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05"; mkdir -p "$O"
cd "$R"
rm -rf /tmp/prof_bench /tmp/prof_calib /tmp/prof_count /tmp/prof_count2 /tmp/prof_normal
timeout -k 10 1000 python3 bench.py --steps 20 --warmup 5 > "$O/bench_line.json" 2> "$O/bench_line.err"
echo "bench done"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python3 "$R/bench.py" --steps 20 --warmup 5 --headline-only > "$O/bench_line_under_rocprof.json" 2> "$O/bench_under_rocprof.err"
cp /tmp/prof_bench/b_kernel_stats.csv "$O/kernel_stats.csv"
echo "bench under rocprof done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_calib -- python3 "$R/scripts/chain_timeline.py" all > "$O/calib_tl.log" 2>&1
python3 "$R/scripts/last_timeline.py" /tmp/prof_calib stats_partial > "$O/calibration_timeline.txt"
echo "calibration timeline done"
PROBE_BATCH_ONLY=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_count -o cb -- python3 "$R/scripts/count_genome_batch_probe.py" 100 all 1 > "$O/count_batch_prof_w1.log" 2>&1
cp /tmp/prof_count/cb_kernel_stats.csv "$O/count_path_batch_kernel_stats.csv"
PROBE_BATCH_ONLY=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_count2 -o cb -- python3 "$R/scripts/count_genome_batch_probe.py" 100 all 2 > "$O/count_batch_prof_w2.log" 2>&1
cp /tmp/prof_count2/cb_kernel_stats.csv "$O/count_path_batch_kernel_stats_two_pipelines.csv"
echo "count path done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_normal -o n -- python3 "$R/scripts/normal_probe.py" > "$O/normal_probe.txt" 2>&1
cp /tmp/prof_normal/n_kernel_stats.csv "$O/normal_kernel_stats.csv"
tail -n 2 "$O/calibration_timeline.txt"; grep "^batch" "$O/count_batch_prof_w1.log" "$O/count_batch_prof_w2.log"; tail -n 4 "$O/normal_probe.txt"; head -c 400 "$O/bench_line.json"
