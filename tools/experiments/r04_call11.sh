#!/bin/bash
# round 4, GPU call 11: GraphedRender (small batches from a HIP graph), kernel-level tests of the colour branch's weight-gradient jobs
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graphed_render" 2>&1 | grep -v amdgpu.ids | tail -n 12
python -m pytest tests/test_training.py -x -q -m gpu -k "wgrad" 2>&1 | grep -v amdgpu.ids | tail -n 6
python tools/small_batch_bench.py bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_small_batch.txt
