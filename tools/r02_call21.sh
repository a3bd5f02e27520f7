#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_stderr.log; tail -c 3000 gpurun_out/r02_bench_line.json
