#!/bin/bash
# round 4, GPU call 35: is the cheaper compositing of the instrumented build a scheduling effect?  Scheduling fences at the
# phase boundaries of composite_ray instead of the timers
set -u
mkdir -p gpurun_out
probe() { echo "## $1"; NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe shipped ""; probe timers-inside-one-instantiation _e5; probe fences _e6; probe fences-one-instantiation _e7; } | tee gpurun_out/r04_x16_timing7.txt
