# The two PMC passes behind bench.py's `roofline.traffic` (profiles/README.md); run on the GPU box, results under gpurun_out/r03/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
rm -rf gpurun_out/r03/pmc_fetch gpurun_out/r03/pmc_write
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_fetch -- python3 scripts/pmc_median.py > gpurun_out/r03/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_write -- python3 scripts/pmc_median.py > gpurun_out/r03/pmc_write.log 2>&1
f=$(ls gpurun_out/r03/pmc_fetch/*/*counter_collection.csv | head -1)
w=$(ls gpurun_out/r03/pmc_write/*/*counter_collection.csv | head -1)
python scripts/pmc_derive.py "$f" "$w" r03
cp profiles/r03_pmc_fetch_size.csv profiles/r03_pmc_write_size.csv profiles/r03_pmc_median.json gpurun_out/r03/
rm -rf gpurun_out/r03/pmc_fetch gpurun_out/r03/pmc_write
cat profiles/r03_pmc_median.json
