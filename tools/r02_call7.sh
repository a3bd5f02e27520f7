#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -q -m gpu -s > gpurun_out/c7_training_tests.log 2>&1
grep -E "passed|failed|Error|assert |worst|native training" gpurun_out/c7_training_tests.log | cut -c1-400 | tail -20
timeout 300 python tools/train_step_breakdown.py bf16 2>&1 | grep -E "per step|free-running"
timeout 300 python tools/train_step_breakdown.py f32 2>&1 | grep -E "per step|free-running"
timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_training.py > gpurun_out/c7_gpu_tests.log 2>&1
tail -5 gpurun_out/c7_gpu_tests.log
timeout 400 python bench.py --train-step > gpurun_out/c7_bench.log 2>&1
tail -1 gpurun_out/c7_bench.log | cut -c1-3000
