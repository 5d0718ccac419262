#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04x
IDV_WINO_MIN_COUT=2 python tests/tools/wino_layers_probe.py 64 2>&1 | grep -v amdgpu.ids | sed 's/^/[mincout 2: dec4 on 144] /' | tee -a gpurun_out/r04x/wino_cfgs.log
