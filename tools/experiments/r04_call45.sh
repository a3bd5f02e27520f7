#!/bin/bash
# round 4, GPU call 45: how much of "points + encoding" is the points' HBM latency?  (timing build with the loads replaced by constants; wrong results)
set -u
mkdir -p gpurun_out
probe() { echo "## $1"; NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 60 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe shipped ""; probe no-point-loads _nl; } | tee gpurun_out/r04_x16_timing13.txt
