#!/bin/bash
# Round-2 profile artefacts:  bash profiles/tools/collect_r02.sh <tag>   (repo root, on a GPU box)
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (fp32 headline + bf16x3 alt record)
#   2. --stats of the fp32 headline alone and of the bf16x3 mode alone (one stream each)
#   3. --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, for both modes  -> r02_traffic_f32.json / r02_traffic_bf16x3.json
#   4. --stats of the DCCRN-CL train step (forward + loss + backward + Adam, B = 32), fp32 and bf16x3 training mode
#   5. --stats of the VAE workloads (NSVAE encoders + KL, two-phase decoder forward, NSVAE train step), B = 32, bf16x3
# Counter passes carry only --kernel-trace (no sys/hip/hsa trace domains).  Outputs land in gpurun_out/prof_<tag>/.
set -o pipefail
tag=${1:-r02}
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
rm -rf "$out" && mkdir -p "$out"
run() {  # name, timeout, rocprof args..., -- bench args
  local name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" > "$out/$name.json" 2> "$out/$name.err" || { echo "FAILED $name"; tail -5 "$out/$name.err"; exit 1; }
  echo "$name done"
}
S="--kernel-trace --stats --output-format csv"
run bench_default $S -d "$out/stats_default" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
run bench_f32 $S -d "$out/stats_f32" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt
run bench_bf16x3 $S -d "$out/stats_bf16x3" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision bf16x3
run bench_train $S -d "$out/stats_train" -- python3 bench.py --workload dccrn_cl_train --steps 3 --warmup 1 --no-cpu-baseline
run bench_train_bf16x3 $S -d "$out/stats_train_bf16x3" -- python3 bench.py --workload dccrn_cl_train --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline
run bench_nsvae_kl $S -d "$out/stats_nsvae_kl" -- python3 bench.py --workload nsvae_kl --precision bf16x3 --batch 32 --steps 4 --warmup 2 --no-cpu-baseline
run bench_twophase $S -d "$out/stats_twophase" -- python3 bench.py --workload twophase --precision bf16x3 --batch 32 --steps 4 --warmup 2 --no-cpu-baseline
run bench_nsvae_train_bf16x3 $S -d "$out/stats_nsvae_train_bf16x3" -- python3 bench.py --workload nsvae_train --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline
P="--kernel-trace --output-format csv"
for mode in f32 bf16x3; do
  prec=$([ $mode = f32 ] && echo fp32 || echo bf16x3)
  run pmc_fetch_$mode $P --pmc FETCH_SIZE -d "$out/pmc_fetch_$mode" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision $prec
  run pmc_write_$mode $P --pmc WRITE_SIZE -d "$out/pmc_write_$mode" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision $prec
done
python3 profiles/tools/summarise_r02.py "$out"
