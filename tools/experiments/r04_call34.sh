#!/bin/bash
# round 4, GPU call 34: phases inside the fused compositing of the x16 kernel (throwaway instrumentation of composite_ray:
# [1] loads [2] alpha [3] prefix scan [5] weights + sigmoids [6] reductions + stores; [4] the whole compositing slot; per iteration, 1 ray per 3 iterations)
set -u
mkdir -p gpurun_out
NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing_e5.so timeout 200 python tools/timing_probe.py --x16 --raw 2>&1 | grep "wave 0" | tail -1 | tee gpurun_out/r04_x16_timing6.txt
