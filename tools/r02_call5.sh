#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -q -m gpu -s > gpurun_out/c5_training_tests.log 2>&1
grep -E "passed|failed|Error|assert |^\[grad" gpurun_out/c5_training_tests.log | cut -c1-400 | tail -30
timeout 300 python tools/train_step_breakdown.py bf16 > gpurun_out/c5_train_breakdown.log 2>&1
grep -E "per step|free-running" gpurun_out/c5_train_breakdown.log
tail -30 gpurun_out/c5_train_breakdown.log | cut -c1-70,180-230 | head -28
timeout 300 python tools/train_step_breakdown.py f32 2>&1 | grep -E "per step|free-running"
