#!/bin/bash
# round 4, GPU call 32: phase breakdown of the coarse pass' kernel (fused bender + trunk, two 32-sample blocks per wave)
set -u
mkdir -p gpurun_out
NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timingc.so timeout 200 python tools/timing_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_coarse_timing.txt
