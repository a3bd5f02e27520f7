#!/bin/bash
# round 4, GPU call 25: x16 kernel variants (epilogue pinned behind the k-step it belongs to; LDS prefetch depth 4 / 6 / 8) against the
# 32x32x16 kernel, A/B in one session
set -u
mkdir -p gpurun_out
B="--no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 --steps 10 --warmup 3"
run() { NRNERF_X16=$2 NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 300 python bench.py $B 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lib[$1] X16=$2', d['value'], d['ms_per_step'], r['frac'], r['kernels_ms_per_step'])" || echo "lib[$1] X16=$2 FAILED"; }
for rep in 1 2; do
run "" 0
run "" 1
run _pf6 1
run _pf8 1
done | tee gpurun_out/r04_x16_ab2.txt
