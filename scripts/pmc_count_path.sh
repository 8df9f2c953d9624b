# Round 5: HBM-side bytes of one count-path call on the K = 100 genome, kernel by kernel -- two PMC passes (FETCH_SIZE, WRITE_SIZE:
# they do not fit one) over scripts/count_genome_batch_probe.py with one pipeline; units and the gfx950 correction as in
# scripts/pmc_derive.py (KiB; FETCH_SIZE x 2 for coalesced streaming reads).  Results: profiles/r05_pmc_count_path.json
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
rm -rf /tmp/pmc_cp_fetch /tmp/pmc_cp_write
export PROBE_BATCH_ONLY=1 PROBE_REPS=2
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_cp_fetch -- python3 scripts/count_genome_batch_probe.py 100 all 1 > $O/pmc_cp_fetch.log 2>&1
echo "fetch pass done"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_cp_write -- python3 scripts/count_genome_batch_probe.py 100 all 1 > $O/pmc_cp_write.log 2>&1
echo "write pass done"
f=$(ls /tmp/pmc_cp_fetch/*/*counter_collection.csv | head -1)
w=$(ls /tmp/pmc_cp_write/*/*counter_collection.csv | head -1)
python3 scripts/pmc_count_path_derive.py "$f" "$w" > $O/pmc_count_path.json
cat $O/pmc_count_path.json
