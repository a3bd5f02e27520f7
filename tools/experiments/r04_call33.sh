#!/bin/bash
# round 4, GPU call 33: x16 iteration breakdown with one compositing instantiation in the loop body instead of four behind a switch
set -u
mkdir -p gpurun_out
probe() { echo "## $1 ${3:-}"; env ${3:-X=1} NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe four-instantiations ""; probe one-instantiation _e5; } | tee gpurun_out/r04_x16_timing5.txt
