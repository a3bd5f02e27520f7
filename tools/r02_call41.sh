#!/bin/bash
NRNERF_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --scaling strong --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 | cut -c1-700
