#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/c27_tests.log 2>&1; tail -8 gpurun_out/c27_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
