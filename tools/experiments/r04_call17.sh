#!/bin/bash
# round 4, GPU call 17: bend_wgrad16 with 512 registers (one wave per SIMD, no scratch) against the shipped two waves + 7 spilled registers
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "" _bw1; do
  rm -rf /tmp/prof_bw
  NRNERF_LIB=$GRAFT_REPO_ROOT/nonrigid_nerf_amd/lib/libnrnerf_hip$v.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bw -o s -- python $GRAFT_REPO_ROOT/tools/train_step_profile.py 16384 bf16 > /tmp/prof_bw.log 2>&1
  db=$(find /tmp/prof_bw -name "*.db" | head -1)
  echo "variant [$v] $(grep 'ms/step' /tmp/prof_bw.log | tail -1)"
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py "$db" 2>&1 | grep "bend_wgrad16\|bend_bwd\|bend_div" | cut -c1-150
done | tee $GRAFT_REPO_ROOT/gpurun_out/r04_bend_wgrad16_ab.txt
