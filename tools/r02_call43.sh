#!/bin/bash
R=$PWD
mkdir -p gpurun_out/prof_train
cd /tmp && export TMPDIR=/tmp
for n in 1024 16384; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_train/f$n -o f -- python $R/tools/train_step_profile.py $n > $R/gpurun_out/prof_train/pmcf$n.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_train/w$n -o w -- python $R/tools/train_step_profile.py $n > $R/gpurun_out/prof_train/pmcw$n.log 2>&1
  python $R/tools/train_pmc_summary.py $R/gpurun_out/prof_train/f$n $R/gpurun_out/prof_train/w$n $n > $R/gpurun_out/r02_train_pmc_$n.txt 2>&1
  cat $R/gpurun_out/r02_train_pmc_$n.txt
done
rm -rf $R/gpurun_out/prof_train/f* $R/gpurun_out/prof_train/w*
