#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04o
timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -q -m gpu -x -k "wgrad or conv_block_grads" 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "wino" 2>&1 | tail -2
for w in 1 0; do IDV_WGRAD_WINO=$w python tests/tools/wgrad_layers_probe.py 32 2>&1 | grep -v amdgpu.ids | sed "s/^/[wino=$w] /" | tee -a gpurun_out/r04o/wgrad_layers.log; done
