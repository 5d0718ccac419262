#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04n
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "wino or gauss" 2>&1 | tail -2
python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04n/wino_cfgs.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r04n/bench_f32.json 2> gpurun_out/r04n/bench_f32.err; python3 -c "
import json;d=json.load(open('gpurun_out/r04n/bench_f32.json'));print(d['value'],d['ms_per_step'],d['roofline']['traffic'])"
