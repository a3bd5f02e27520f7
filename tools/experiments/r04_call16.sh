#!/bin/bash
# round 4, GPU call 16: device-side re-pack of all images in one launch (repack_batch_kernel): parity of the refresh, step times
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_training.py -q -m gpu -k "weight_refresh or refreshes_weights or optimiser_step_of_any_kind or graphed" 2>&1 | grep -v amdgpu.ids | tail -n 8
python tools/train_step_scaling.py 2>&1 | grep -v amdgpu.ids | grep "bf16" | tee gpurun_out/r04_train_step_modes_batched_repack.txt
