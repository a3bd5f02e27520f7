#!/bin/bash
# round 4, GPU call 36: x16 with global (not flat) accesses in composite_ray and the ring tail issued after the compositing
set -u
mkdir -p gpurun_out
probe() { echo "## $1"; NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 200 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe global-accesses+tail-after ""; probe same-rebuilt _e8; probe same-one-instantiation-with-timers _e9; } | tee gpurun_out/r04_x16_timing8.txt
