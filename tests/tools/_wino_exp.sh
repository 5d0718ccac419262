#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04p
timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -q -m gpu -x -k "wgrad or conv_block_grads" 2>&1 | tail -2
IDV_WGRAD_WINO_JT=32 timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -q -m gpu -x -k "wgrad or conv_block_grads" 2>&1 | tail -2
for jt in 16 32; do IDV_WGRAD_WINO_JT=$jt python tests/tools/wgrad_layers_probe.py 32 2>&1 | grep -v amdgpu.ids | sed "s/^/[jt=$jt] /" | tee -a gpurun_out/r04p/wgrad_layers.log; done
