#!/bin/bash
# round 4, GPU call 37: x16 kernel specialised per compositing case (one composite_ray instantiation, fenced), biases in the counted
# queue, global accesses in composite_ray: iteration breakdown, bench, and the parity tests that cover the changed code
set -u
mkdir -p gpurun_out
{ echo "## per-case kernel (129..192 samples), timing build"; NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; } | tee gpurun_out/r04_x16_timing9.txt
B="--no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 --steps 10 --warmup 3"
run() { NRNERF_X16=$1 timeout 300 python bench.py $B 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('X16=$1', d['value'], d['ms_per_step'], r['frac'], r['kernels_ms_per_step'])" || echo "X16=$1 FAILED"; }
{ run 1; run 0; run 1; } | tee gpurun_out/r04_x16_ab5.txt
python -m pytest tests/test_gpu_parity.py tests/test_fitted_checkpoint.py -q -m gpu -x -k "fused_into_the_network or fitted or split_bender or chunk or golden or surface" 2>&1 | grep -v amdgpu.ids | tail -n 8
