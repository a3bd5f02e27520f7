#!/bin/bash
# round 4, GPU call 22: training with more than 256 samples per pass (composite backward up to 16 samples per lane, the fine pass bends all
# merged samples beyond the split path's 8-bit ranks); boundary contract test
set -u
mkdir -p gpurun_out
python -m pytest tests/test_training.py tests/test_gpu_parity.py -q -m gpu -k "350_samples or boundary_contract or composite" 2>&1 | grep -v amdgpu.ids | tail -n 15
