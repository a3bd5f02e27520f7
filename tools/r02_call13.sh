#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -q -m gpu -s -k "native_bender" > gpurun_out/c13_tests.log 2>&1; tail -5 gpurun_out/c13_tests.log; grep -n "^\[" gpurun_out/c13_tests.log
timeout 400 python tools/train_step_breakdown.py bf16 1024 > gpurun_out/c13_breakdown_1024.log 2>&1; head -60 gpurun_out/c13_breakdown_1024.log | cut -c1-200
timeout 400 python tools/train_step_scaling.py 2>&1 | grep bf16
