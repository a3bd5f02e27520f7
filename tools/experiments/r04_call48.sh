#!/bin/bash
# round 4, GPU call 48 (the last GPU seconds; probe only, nothing shipped): the fused compositing without its one unconditional
# `s_waitcnt vmcnt(0)` -- hipcc puts it at the join behind the conditional near / far load of `if (!a.z)`, where it also waits for
# whatever else is in the vector-memory queue (the ring's LDS-DMA, the next iteration's points); a fused final pass always has depths
set -u
mkdir -p gpurun_out
probe() { echo "## $1"; NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing$2.so timeout 30 python tools/timing_probe.py --x16 2>&1 | grep "wave 0" | tail -1; }
{ probe shipped ""; probe depths-known-at-compile-time _hz; } | tee gpurun_out/r04_x16_timing14.txt
