#!/bin/bash
# round 4, GPU call 30: x16 iteration breakdown with the loop's branches hinted (the non-compositing iteration falls through to the back edge)
set -u
mkdir -p gpurun_out
probe() { echo "## $1 ${3:-}"; env ${3:-X=1} NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe before ""; probe hinted _e3; probe hinted-unfused _e3 NRNERF_UNFUSED_COMPOSITE=1; } | tee gpurun_out/r04_x16_timing3.txt
