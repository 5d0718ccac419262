#!/bin/bash
# Round-4 profile artefacts:  bash profiles/tools/collect_r04.sh <tag>   (repo root, on a GPU box)
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (fp32 headline + bf16x3 alt record + two-stream checks)
#   2. --stats of the fp32 headline alone (the launches `roofline` describes) and of the bf16x3 mode alone
#   3. --stats of the fp32 workloads of BASELINE configs 2 / 3 / 5 (cvae_elbo B = 64 x 5 samples, nsvae_kl, twophase: B = 32) and of
#      the fp32 train steps (dccrn_cl_train, nsvae_train, twophase_train: B = 32)
#   4. --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, both modes  -> r04_traffic_f32.json / r04_traffic_bf16x3.json
# Counter passes carry only --kernel-trace (no sys / hip / hsa trace domains); the program goes directly after `--`.
set -o pipefail
tag=${1:-r04}
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
rm -rf "$out" && mkdir -p "$out"
run() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" > "$out/$name.json" 2> "$out/$name.err" || { echo "FAILED $name"; tail -5 "$out/$name.err"; exit 1; }
  echo "$name done"
}
S="--kernel-trace --stats --output-format csv"
run bench_default $S -d "$out/stats_default" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
run bench_f32 $S -d "$out/stats_f32" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt
run bench_bf16x3 $S -d "$out/stats_bf16x3" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision bf16x3
run bench_cvae_f32 $S -d "$out/stats_cvae_f32" -- python3 bench.py --workload cvae_elbo --steps 2 --warmup 1 --no-cpu-baseline
run bench_nsvae_kl_f32 $S -d "$out/stats_nsvae_kl_f32" -- python3 bench.py --workload nsvae_kl --batch 32 --steps 4 --warmup 2 --no-cpu-baseline
run bench_twophase_f32 $S -d "$out/stats_twophase_f32" -- python3 bench.py --workload twophase --batch 32 --steps 4 --warmup 2 --no-cpu-baseline
run bench_enhance_f32 $S -d "$out/stats_enhance_f32" -- python3 bench.py --workload enhance --steps 3 --warmup 1 --no-cpu-baseline
run bench_train_f32 $S -d "$out/stats_train_f32" -- python3 bench.py --workload dccrn_cl_train --steps 3 --warmup 1 --no-cpu-baseline
run bench_nsvae_train_f32 $S -d "$out/stats_nsvae_train_f32" -- python3 bench.py --workload nsvae_train --steps 3 --warmup 1 --no-cpu-baseline
run bench_twophase_train_f32 $S -d "$out/stats_twophase_train_f32" -- python3 bench.py --workload twophase_train --steps 3 --warmup 1 --no-cpu-baseline
P="--kernel-trace --output-format csv"
for mode in f32 bf16x3; do
  prec=$([ $mode = f32 ] && echo fp32 || echo bf16x3)
  run pmc_fetch_$mode $P --pmc FETCH_SIZE -d "$out/pmc_fetch_$mode" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision $prec
  run pmc_write_$mode $P --pmc WRITE_SIZE -d "$out/pmc_write_$mode" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision $prec
done
python3 profiles/tools/summarise_r04.py "$out"
