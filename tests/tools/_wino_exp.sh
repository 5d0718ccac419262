#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04t
IDV_WINO_CONV_MINC=64 python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | sed 's/^/[minc=64] /' | tee -a gpurun_out/r04t/wino_cfgs.log
python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04t/wino_cfgs.log
