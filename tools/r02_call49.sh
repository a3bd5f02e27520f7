#!/bin/bash
R=$PWD
mkdir -p gpurun_out/prof_train
cd /tmp && export TMPDIR=/tmp
for V in "" _S2 _S8; do
  export NRNERF_LIB=$R/nonrigid_nerf_amd/lib/libnrnerf_hip$V.so
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train/s$V -o s -- python $R/tools/train_step_profile.py 16384 > $R/gpurun_out/prof_train/stats$V.log 2>&1
  DB=$(find $R/gpurun_out/prof_train/s$V -name "*.db" | head -1)
  echo "variant [$V]"; python $R/tools/rocprof_summary.py "$DB" 2>&1 | grep "trunk_wgrad" | cut -c1-80
done
find $R/gpurun_out/prof_train -name "*.db" -delete
