#!/bin/bash
# round 4, GPU call 38: what made the instrumented compositing faster?  One instruction at its start: s_memtime / a full wait / an LDS wait
set -u
mkdir -p gpurun_out
probe() { echo "## $1"; NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe plain ""; probe s_memtime _x5; probe wait-all _x6; probe wait-lds _x7; } | tee gpurun_out/r04_x16_timing10.txt
