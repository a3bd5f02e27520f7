#!/bin/bash
bash tools/collect_profiles.sh r02 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_stderr.log; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['train_step']['ms_per_step'], d['psnr_vs_oracle_db']['rgb_map'], d['roofline']['library_gemm_tflops_same_box'])"
