#!/bin/bash
# round 4, GPU call 26: x16 kernel with the compositing fused in: bit identity against the composite kernel, accuracy on the fitted
# checkpoints, A/B against the 32x32x16 kernel
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_fitted_checkpoint.py -q -m gpu -k "fused_into_the_network or fitted or bf16 or split_bender or large or chunk" 2>&1 | grep -v amdgpu.ids | tail -n 12
B="--no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 --steps 10 --warmup 3"
run() { NRNERF_X16=$1 timeout 300 python bench.py $B 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('X16=$1', d['value'], d['ms_per_step'], r['frac'], r['kernels_ms_per_step'])" || echo "X16=$1 FAILED"; }
for rep in 1 2; do run 0; run 1; done | tee gpurun_out/r04_x16_ab3.txt
