#!/bin/bash
R=$PWD
mkdir -p gpurun_out/prof_train
timeout 300 python -m pytest tests/test_training.py -q -m gpu -k "wgrad" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_train/f0 -o f -- python $R/tools/train_step_profile.py 16384 > $R/gpurun_out/prof_train/pmcf0.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_train/w0 -o w -- python $R/tools/train_step_profile.py 16384 > $R/gpurun_out/prof_train/pmcw0.log 2>&1
python $R/tools/train_pmc_summary.py $R/gpurun_out/prof_train/f0 $R/gpurun_out/prof_train/w0 16384 | grep "trunk_wgrad\|kernel "
timeout 200 python $R/tools/train_step_scaling.py 2>&1 | grep "bf16"
rm -rf $R/gpurun_out/prof_train/f* $R/gpurun_out/prof_train/w*
