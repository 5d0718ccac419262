#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04g
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r04g/bench_f32_wino.json 2> gpurun_out/r04g/bench_f32_wino.err; python3 -c "
import json;d=json.load(open('gpurun_out/r04g/bench_f32_wino.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel'][:80]);[print(k[:110],v) for k,v in d['roofline']['per_kernel'].items()]"
IDV_WINO=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r04g/bench_f32_nowino.json 2>/dev/null; python3 -c "
import json;d=json.load(open('gpurun_out/r04g/bench_f32_nowino.json'));print('IDV_WINO=0',d['value'],d['ms_per_step'])"
python -m pytest tests/test_gpu_models.py tests/test_gpu_batch_sizes.py -q -m gpu -x 2>&1 | tail -3
