#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/debug_grads.py > gpurun_out/c4_debug_grads.log 2>&1
cat gpurun_out/c4_debug_grads.log | tail -80
timeout 900 python -m pytest tests/test_training.py -q -m gpu > gpurun_out/c4_training_tests.log 2>&1
grep -E "passed|failed|Error|assert " gpurun_out/c4_training_tests.log | tail -30
timeout 300 python tools/train_step_breakdown.py bf16 > gpurun_out/c4_train_breakdown.log 2>&1
tail -45 gpurun_out/c4_train_breakdown.log | cut -c1-220
