#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04w
python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04w/wino_cfgs.log
IDV_WINO_XCD_SPLIT=1 python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | sed 's/^/[xsplit 418] /' | tee -a gpurun_out/r04w/wino_cfgs.log
IDV_WINO_XCD_SPLIT=1 IDV_WINO_CFG=228 python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | sed 's/^/[xsplit 228] /' | tee -a gpurun_out/r04w/wino_cfgs.log
IDV_WINO_XCD_SPLIT=0 IDV_WINO_CFG=228 python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | sed 's/^/[nosplit 228] /' | tee -a gpurun_out/r04w/wino_cfgs.log
IDV_WINO_XCD_SPLIT=1 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "wino" 2>&1 | tail -2
