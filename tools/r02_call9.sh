#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/c9_gpu_tests.log 2>&1
tail -6 gpurun_out/c9_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 500 python bench.py > gpurun_out/c9_bench.log 2>&1
tail -1 gpurun_out/c9_bench.log | cut -c1-3500
bash tools/collect_profiles.sh r02b > gpurun_out/collect_r02b.log 2>&1
tail -3 gpurun_out/collect_r02b.log | cut -c1-300
head -8 gpurun_out/r02b_kernel_stats.txt | cut -c1-200
