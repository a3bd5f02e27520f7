#!/bin/bash
# round 4, GPU call 19: training with exact (Jacobian) view directions: reference-autograd golden, oracle autograd on the device
set -u
mkdir -p gpurun_out
python -m pytest tests/test_training.py -q -m gpu -k "exact_viewdirs or divergence or full_training_iteration" 2>&1 | grep -v amdgpu.ids | tail -n 25
