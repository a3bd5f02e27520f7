#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/c36_tests.log 2>&1; tail -4 gpurun_out/c36_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/collect_profiles.sh r02 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_stderr.log; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['train_step'], d['psnr_vs_oracle_db']['rgb_map'], d['roofline']['library_gemm_tflops_same_box'])"
