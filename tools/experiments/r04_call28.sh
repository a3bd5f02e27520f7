#!/bin/bash
# round 4, GPU call 28: where an iteration of the 16x16x32 kernel spends its cycles, and at which clock (NRN_TIMING build of nrnerf_net_x16.hip)
set -u
mkdir -p gpurun_out
NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing.so timeout 300 python tools/timing_probe.py --x16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_x16_timing.txt
