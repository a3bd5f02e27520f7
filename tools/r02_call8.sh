#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fitted_checkpoint.py -q -m gpu -s > gpurun_out/c8_fitted.log 2>&1
grep -E "passed|failed|PSNR|assert" gpurun_out/c8_fitted.log | cut -c1-600
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "16bit or split_bender or viewdirs_fp32 or repeated or device_side" > gpurun_out/c8_parity16.log 2>&1
grep -E "passed|failed|^\[|assert |AssertionError" gpurun_out/c8_parity16.log | cut -c1-330 | tail -60
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"; }
for i in 1 2; do timeout 200 $B 2>&1 | tail -1 | show "bf16 (single-product bender)"; done
timeout 200 $B --precision f16 2>&1 | tail -1 | show "f16 (split bender)"
NRNERF_SPLIT_COARSE=1 timeout 200 $B 2>&1 | tail -1 | show "bf16 split coarse too"
NRNERF_FUSED_FINE_BENDER=1 timeout 200 $B 2>&1 | tail -1 | show "bf16 fused fine"
timeout 200 $B --scene synthetic 2>&1 | tail -1 | show "bf16 synthetic"
