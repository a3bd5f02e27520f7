#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -q -m gpu -k "wgrad or bf16 or fits" > gpurun_out/c35_tests.log 2>&1; tail -3 gpurun_out/c35_tests.log
timeout 400 python tools/train_step_scaling.py 2>&1 | grep "bf16"
timeout 400 python tools/train_step_breakdown.py bf16 16384 > gpurun_out/c35_breakdown_16384.log 2>&1; grep -n "trunk_wgrad" gpurun_out/c35_breakdown_16384.log | cut -c1-60,150-240
