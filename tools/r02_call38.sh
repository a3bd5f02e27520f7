#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_training.py -q -m gpu -k "wgrad or point_the_same or fits" 2>&1 | tail -3
timeout 300 python tools/train_step_scaling.py 2>&1 | grep "bf16"
NRNERF_WGRAD_DIRECT=1 timeout 300 python tools/train_step_scaling.py 2>&1 | grep "bf16"
