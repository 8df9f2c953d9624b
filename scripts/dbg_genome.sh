export ROCCO_HIP_DEBUG=1
G=${1:-chr1,chr15,chr21}
timeout -k 10 200 python scripts/calib_timeline.py $G 2>&1 | grep -E "lean model|\[round|lean round|host\]|\[search\]|\[pilot\]|adopted" | tail -${2:-70}
