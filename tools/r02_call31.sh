#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py tests/test_distributed.py -q -m gpu > gpurun_out/c31_tests.log 2>&1; tail -3 gpurun_out/c31_tests.log
timeout 400 python tools/train_step_scaling.py 2>&1 | grep "bf16"
timeout 400 python tools/train_step_breakdown.py bf16 1024 > gpurun_out/c31_breakdown_1024.log 2>&1; grep -n "per step\|free-running" gpurun_out/c31_breakdown_1024.log
timeout 400 python tools/train_step_breakdown.py bf16 16384 > gpurun_out/c31_breakdown_16384.log 2>&1; grep -n "per step\|free-running" gpurun_out/c31_breakdown_16384.log
