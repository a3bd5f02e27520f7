#!/bin/bash
R=$PWD
mkdir -p gpurun_out/prof_train
cd /tmp && export TMPDIR=/tmp
for n in 1024 16384; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train/s$n -o s -- python $R/tools/train_step_profile.py $n > $R/gpurun_out/prof_train/stats$n.log 2>&1
  tail -1 $R/gpurun_out/prof_train/stats$n.log
  DB=$(find $R/gpurun_out/prof_train/s$n -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" 2>&1 | cut -c1-230 | head -24 > $R/gpurun_out/r02_train_kernel_stats_$n.txt
  head -16 $R/gpurun_out/r02_train_kernel_stats_$n.txt | cut -c1-200
done
find $R/gpurun_out/prof_train -name "*.db" -delete
