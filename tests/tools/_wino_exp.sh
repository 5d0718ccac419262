#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04i
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "wino or gauss" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_models.py tests/test_gpu_batch_sizes.py -q -m gpu -x 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r04i/bench_f32.json 2> gpurun_out/r04i/bench_f32.err; python3 -c "
import json;d=json.load(open('gpurun_out/r04i/bench_f32.json'));print(d['value'],d['ms_per_step']);[print(k[:120],v) for k,v in d['roofline']['per_kernel'].items()]"
for w in dccrn_cl_train nsvae_train twophase_train; do python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04i/bench_$w.json 2>/dev/null; python3 -c "
import json,sys;d=json.load(open('gpurun_out/r04i/bench_$w.json'));print('$w',d['value'],d['ms_per_step'])"; done
