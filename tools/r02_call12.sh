#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -q -m gpu -s > gpurun_out/c12_tests.log 2>&1; tail -5 gpurun_out/c12_tests.log; grep -n "^\[" gpurun_out/c12_tests.log
timeout 400 python tools/train_step_breakdown.py bf16 16384 > gpurun_out/c12_breakdown_16384.log 2>&1; head -60 gpurun_out/c12_breakdown_16384.log | cut -c1-200
