#!/bin/bash
R=$PWD
mkdir -p gpurun_out/prof_train
NRNERF_WGRAD_RING=1 timeout 300 python -m pytest tests/test_training.py -q -m gpu -k "wgrad or point_the_same" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for ring in 0 1; do
  export NRNERF_WGRAD_RING=$ring
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_train/f$ring -o f -- python $R/tools/train_step_profile.py 16384 > $R/gpurun_out/prof_train/pmcf$ring.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_train/w$ring -o w -- python $R/tools/train_step_profile.py 16384 > $R/gpurun_out/prof_train/pmcw$ring.log 2>&1
  echo "ring=$ring"; python $R/tools/train_pmc_summary.py $R/gpurun_out/prof_train/f$ring $R/gpurun_out/prof_train/w$ring 16384 | grep "trunk_wgrad\|kernel "
  timeout 200 python $R/tools/train_step_scaling.py 2>&1 | grep "bf16"
done
rm -rf $R/gpurun_out/prof_train/f* $R/gpurun_out/prof_train/w*
