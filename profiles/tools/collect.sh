#!/bin/bash
# Re-collects the judged profile artefacts on a GPU box:  bash profiles/tools/collect.sh <tag>   (from the repo root)
#   1. rocprofv3 --kernel-trace --stats of the default bench command (2 sub-batch streams)
#   2. the same with IDV_STREAM_SPLIT=1 (the single-stream launches bench.py's roofline block describes)
#   3. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, single stream  -> profiles/r01_traffic.json
# Counter passes carry only --kernel-trace (no sys/hip/hsa trace domains).  Outputs land in gpurun_out/prof_<tag>/.
set -o pipefail
tag=${1:-run}
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
rm -rf "$out" && mkdir -p "$out"
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_default" -- $B > "$out/bench_default.json" 2> "$out/bench_default.err" || exit 1
echo "stats default done"
export IDV_STREAM_SPLIT=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_single" -- $B > "$out/bench_single.json" 2> "$out/bench_single.err" || exit 1
echo "stats single-stream done"
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $P > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err" || exit 1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- $P > "$out/pmc_write.json" 2> "$out/pmc_write.err" || exit 1
echo "pmc write done"
python3 profiles/tools/summarise.py "$out"
