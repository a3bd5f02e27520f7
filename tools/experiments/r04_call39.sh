#!/bin/bash
# round 4, GPU call 39: the timers inside composite_ray without their bookkeeping: s_memtime + lgkmcnt(0) / lgkmcnt(0) alone at the phase boundaries
set -u
mkdir -p gpurun_out
probe() { echo "## $1"; NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe s_memtime+wait-at-phase-boundaries _f1; probe lds-wait-at-phase-boundaries _f2; } | tee gpurun_out/r04_x16_timing11.txt
