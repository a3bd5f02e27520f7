#!/bin/bash
# Build nonrigid_nerf_amd/lib/libnrnerf_hip_timing_bend<suffix>.so: the shipped objects with nrnerf_bend_x16.hip rebuilt with -DNRN_TIMING
# (phase cycle counters of bend_kernel_x16; tools/timing_probe_bender.py).    tools/build_timing_bender.sh [suffix] [extra hipcc flags ...]
set -euo pipefail
cd "$(dirname "$0")/../nonrigid_nerf_amd/csrc"
SUF=${1:-}; [ $# -gt 0 ] && shift
TMP=$(mktemp -d)
FL=$(make -n -B build/nrnerf_bend_x16.o 2>/dev/null | grep hipcc | sed 's/ -c .*//')
$FL -DNRN_TIMING "$@" -c nrnerf_bend_x16.hip -o "$TMP/nrnerf_bend_x16.o"
LINK=$(make -n -B ../lib/libnrnerf_hip.so 2>/dev/null | grep -- "-shared" | sed "s#build/nrnerf_bend_x16.o#$TMP/nrnerf_bend_x16.o#; s#../lib/libnrnerf_hip.so#../lib/libnrnerf_hip_timing_bend$SUF.so#")
eval "$LINK"
rm -rf "$TMP"
echo "built nonrigid_nerf_amd/lib/libnrnerf_hip_timing_bend$SUF.so"
