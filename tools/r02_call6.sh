#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/debug_train_steps.py bf16 > gpurun_out/c6_debug_steps.log 2>&1; tail -40 gpurun_out/c6_debug_steps.log
timeout 300 python tools/debug_train_steps.py f32 2>&1 | tail -14
timeout 900 python -m pytest tests/test_training.py -q -m gpu -s > gpurun_out/c6_training_tests.log 2>&1
grep -E "passed|failed|Error|assert |worst" gpurun_out/c6_training_tests.log | cut -c1-400 | tail -20
