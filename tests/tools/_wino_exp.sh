#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04l
python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04l/wino_cfgs.log
IDV_WINO_PH0=3 python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | sed 's/^/[ph0=3] /' | tee -a gpurun_out/r04l/wino_cfgs.log
IDV_WINO_PH0=3 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "wino" 2>&1 | tail -2
