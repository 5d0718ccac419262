#!/bin/bash
# Round-3 workload lines (one bench.py run each, outputs under gpurun_out/meas_r03/):  bash profiles/tools/measure_r03.sh [fp32|all]
set -o pipefail
o=gpurun_out/meas_r03; mkdir -p $o
b() { name=$1; shift; timeout -k 10 280 python bench.py "$@" --no-cpu-baseline --no-alt > $o/$name.json 2> $o/$name.err || { echo FAIL $name; tail -3 $o/$name.err; exit 1; }; python3 - <<PY
import json
d = json.load(open("$o/$name.json"))
print("$name", d["value"], "utt/s", d["ms_per_step"], "ms/step")
PY
}
b nsvae_kl_f32 --workload nsvae_kl --batch 32 --steps 8 --warmup 2
b twophase_f32 --workload twophase --batch 32 --steps 8 --warmup 2
b cvae_f32 --workload cvae_elbo --steps 4 --warmup 2
b enhance_f32 --workload enhance --steps 5 --warmup 2
b enhance_cmask_f32 --workload enhance_complex_mask --steps 5 --warmup 2
b dccrn_cl_train_f32 --workload dccrn_cl_train --steps 5 --warmup 2
b nsvae_train_f32 --workload nsvae_train --steps 5 --warmup 2
b twophase_train_f32 --workload twophase_train --steps 5 --warmup 2
b cvae_train_f32 --workload cvae_train --steps 4 --warmup 2
if [ "${1:-fp32}" = all ]; then
  b nsvae_kl_bf16 --workload nsvae_kl --precision bf16x3 --batch 32 --steps 10 --warmup 3
  b twophase_bf16 --workload twophase --precision bf16x3 --batch 32 --steps 10 --warmup 3
  b cvae_bf16 --workload cvae_elbo --precision bf16x3 --steps 8 --warmup 2
  b enhance_bf16 --workload enhance --precision bf16x3 --steps 5 --warmup 2
  b dccrn_cl_train_bf16 --workload dccrn_cl_train --precision bf16x3 --steps 5 --warmup 2
  b nsvae_train_bf16 --workload nsvae_train --precision bf16x3 --steps 6 --warmup 2
  b twophase_train_bf16 --workload twophase_train --precision bf16x3 --steps 6 --warmup 2
  b cvae_train_bf16 --workload cvae_train --precision bf16x3 --steps 4 --warmup 2
fi
