#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_distributed.py -q -m gpu -k "data_parallel" > gpurun_out/c17_tests.log 2>&1; tail -30 gpurun_out/c17_tests.log
