#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py tests/test_distributed.py -q -m gpu > gpurun_out/c32_tests.log 2>&1; tail -3 gpurun_out/c32_tests.log
timeout 400 python tools/train_step_scaling.py 2>&1 | grep "bf16"
