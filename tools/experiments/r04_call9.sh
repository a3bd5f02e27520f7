#!/bin/bash
# round 4, GPU call 9: the fp32 mode's native weight-gradient kernel (trunk_wgrad_f32): kernel test, the gradient-parity tests of
# the fp32 mode, the step times of both modes (profiles/r03_train_step_modes.txt has the library-GEMM numbers: fp32 16 384 rays 97.4 ms)
set -u
mkdir -p gpurun_out
python -m pytest tests/test_training.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -n 15 > gpurun_out/r04_c9_train_tests.log
tail -n 4 gpurun_out/r04_c9_train_tests.log
python tools/train_step_scaling.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_train_step_modes.txt
cat gpurun_out/r04_train_step_modes.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f32
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o s -- python $GRAFT_REPO_ROOT/tools/train_step_profile.py 16384 f32 > /tmp/prof_f32.log 2>&1
db=$(find /tmp/prof_f32 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py "$db" 2>&1 | head -20 | cut -c1-220 > $GRAFT_REPO_ROOT/gpurun_out/r04_train_kernel_stats_16384_f32.txt
head -12 $GRAFT_REPO_ROOT/gpurun_out/r04_train_kernel_stats_16384_f32.txt
