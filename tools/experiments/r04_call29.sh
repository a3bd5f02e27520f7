#!/bin/bash
# round 4, GPU call 29: x16 iteration breakdown: shipped code, compositing unfused, no epilogue conversion (upper bound of what hiding
# the epilogue's VALU work can give; wrong results), epilogue not pinned behind its k-step
set -u
mkdir -p gpurun_out
probe() { echo "## $1 ${3:-}"; env ${3:-X=1} NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe shipped ""; probe unfused "" NRNERF_UNFUSED_COMPOSITE=1; probe no-conversion _e1; probe not-pinned _e2; } | tee gpurun_out/r04_x16_timing2.txt
