#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04v
python -m pytest tests -q -m gpu -x > gpurun_out/r04v/full_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r04v/full_suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04v/bench_default.json 2> gpurun_out/r04v/bench_default.err; python3 -c "
import json;d=json.load(open('gpurun_out/r04v/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('frac_executed'),d['alt']['value'],d['two_stream']['value'],d['two_stream']['bit_exact'],d['cpu_baseline']['value'])"
for w in dccrn_cl_train nsvae_train twophase_train cvae_train; do python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04v/bench_$w.json 2>/dev/null; python3 -c "
import json,sys;d=json.load(open('gpurun_out/r04v/bench_$w.json'));print('$w',d['value'],d['ms_per_step'])"; done
