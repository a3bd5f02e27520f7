#!/bin/bash
# Build nonrigid_nerf_amd/lib/libnrnerf_hip_timing<suffix>.so: the shipped objects with the 129..192-sample case of the 16x16x32
# trunk-only kernel (nrnerf_net_x16.hip, -DNRN_X16_EPL=3) rebuilt with -DNRN_TIMING (phase cycle counters, tools/timing_probe.py --x16).
#   tools/build_timing_x16.sh [suffix] [extra hipcc flags ...]        (run `make -C nonrigid_nerf_amd/csrc` first)
#   NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_timing.so python tools/timing_probe.py --x16       (on the GPU box)
# profiles/r04_iteration_breakdown.txt was collected this way (tools/experiments/r04_call28.sh ... r04_call43.sh).
set -euo pipefail
cd "$(dirname "$0")/../nonrigid_nerf_amd/csrc"
SUF=${1:-}; [ $# -gt 0 ] && shift
TMP=$(mktemp -d)
FL=$(make -n -B build/nrnerf_net_x16_e3.o 2>/dev/null | grep hipcc | sed 's/ -c .*//')
$FL -DNRN_TIMING "$@" -c nrnerf_net_x16.hip -o "$TMP/nrnerf_net_x16_e3.o"
LINK=$(make -n -B ../lib/libnrnerf_hip.so 2>/dev/null | grep -- "-shared" | sed "s#build/nrnerf_net_x16_e3.o#$TMP/nrnerf_net_x16_e3.o#; s#../lib/libnrnerf_hip.so#../lib/libnrnerf_hip_timing$SUF.so#")
eval "$LINK"
rm -rf "$TMP"
echo "built nonrigid_nerf_amd/lib/libnrnerf_hip_timing$SUF.so"
