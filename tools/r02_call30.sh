#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py tests/test_c_abi.py -q -m gpu -s > gpurun_out/c30_tests.log 2>&1; tail -5 gpurun_out/c30_tests.log; grep -n "^\[" gpurun_out/c30_tests.log | cut -c1-160 | tail -8
timeout 400 python tools/train_step_scaling.py 2>&1 | grep "bf16"
