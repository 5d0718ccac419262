#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04k
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "wino or gauss" 2>&1 | tail -2
python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04k/wino_cfgs.log
IDV_WINO_HALF=0 python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | sed 's/^/[half=0] /' | tee -a gpurun_out/r04k/wino_cfgs.log
IDV_WINO_PH0=2 python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | sed 's/^/[ph0=2] /' | tee -a gpurun_out/r04k/wino_cfgs.log
IDV_WINO_PH0=2 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "wino" 2>&1 | tail -2
