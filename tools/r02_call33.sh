#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -q -m gpu > gpurun_out/c33_tests.log 2>&1; tail -3 gpurun_out/c33_tests.log
