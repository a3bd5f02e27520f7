#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -q -m gpu -s -k "oracle_autograd" > gpurun_out/c23_tests.log 2>&1; tail -4 gpurun_out/c23_tests.log; grep -n "^\[" gpurun_out/c23_tests.log | cut -c1-200
