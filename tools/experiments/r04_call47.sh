#!/bin/bash
# round 4, GPU call 47: same-box A/B of the x16 kernel with the next iteration's points requested ahead (lib) against the previous build (lib_prev)
set -u
mkdir -p gpurun_out
B="--no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 --steps 10 --warmup 3"
run() { NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 60 python bench.py $B 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lib[$1]', d['value'], d['ms_per_step'], r['frac'], r['kernels_ms_per_step'])" || echo "lib[$1] FAILED"; }
{ run _prev; run ""; run _prev; run ""; } | tee gpurun_out/r04_x16_points_ahead_ab.txt
