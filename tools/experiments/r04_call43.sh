#!/bin/bash
# round 4, GPU call 43: x16 iteration breakdown with ONE compositing copy on the one-ray-per-wave path (the copies for 2 / 4 rays per
# wave behind it), the ring tail back in front of the compositing, optional parts of composite_ray hinted
set -u
mkdir -p gpurun_out
NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing.so timeout 100 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1 | tee gpurun_out/r04_x16_timing12.txt
