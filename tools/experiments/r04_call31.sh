#!/bin/bash
# round 4, GPU call 31: x16 iteration breakdown with gridDim.x held in a register (no scalar load from the dispatch packet in the loop)
set -u
mkdir -p gpurun_out
probe() { echo "## $1 ${3:-}"; env ${3:-X=1} NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe gdim-in-register _e4; probe gdim-in-register-unfused _e4 NRNERF_UNFUSED_COMPOSITE=1; } | tee gpurun_out/r04_x16_timing4.txt
