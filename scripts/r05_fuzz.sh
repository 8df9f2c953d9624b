# Round-5 fuzz tallies on the round's build (GPU against the CPU oracle, bit for bit); results under gpurun_out/r05/
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05"; mkdir -p "$O"
cd "$R"
S=${FUZZ_SECONDS:-300}
timeout -k 10 $((S + 200)) python3 tests/tools/fuzz_batch.py $S > "$O/fuzz_batch.txt" 2>&1; tail -n 1 "$O/fuzz_batch.txt"
timeout -k 10 $((S + 200)) python3 tests/tools/fuzz_parity.py $S > "$O/fuzz_parity.txt" 2>&1; tail -n 1 "$O/fuzz_parity.txt"
timeout -k 10 $((S + 200)) python3 tests/tools/fuzz_count_path.py $S > "$O/fuzz_count_path.txt" 2>&1; tail -n 1 "$O/fuzz_count_path.txt"
