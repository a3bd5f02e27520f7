#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "split_bender_path_equals" > gpurun_out/c25_tests.log 2>&1; tail -12 gpurun_out/c25_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"; }
timeout 200 $B --use-viewdirs 2>&1 | tail -1 | show "viewdirs split"
NRNERF_FUSED_FINE_BENDER=1 timeout 200 $B --use-viewdirs 2>&1 | tail -1 | show "viewdirs fused"
timeout 200 $B --use-viewdirs --bend-depth 7 2>&1 | tail -1 | show "config4 split"
NRNERF_FUSED_FINE_BENDER=1 timeout 200 $B --use-viewdirs --bend-depth 7 2>&1 | tail -1 | show "config4 fused"
