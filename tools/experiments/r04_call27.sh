#!/bin/bash
# round 4, GPU call 27: x16 kernel with the biases travelling in the counted LDS queue (no lgkmcnt(0) per tile pair) against the
# previous build (plain bias loads), plus the split-vs-fused tests restated for the 16x16x32 trunk-only pass
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_fitted_checkpoint.py -q -m gpu -k "fused_into_the_network or fitted or split_bender or chunk" 2>&1 | grep -v amdgpu.ids | tail -n 12
B="--no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 --steps 10 --warmup 3"
run() { NRNERF_X16=$2 NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 300 python bench.py $B 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lib[$1] X16=$2', d['value'], d['ms_per_step'], r['frac'], r['kernels_ms_per_step'])" || echo "lib[$1] X16=$2 FAILED"; }
for rep in 1 2; do
run _prev 1
run "" 1
done | tee gpurun_out/r04_x16_ab4.txt
