#!/bin/bash
# round 4, GPU call 10: the colour branch of the view-dependent head inside the training kernels (trunk_fwd_train / trunk_bwd <.., VIEWS>,
# three more jobs in trunk_wgrad): the gradient tests, then the view-dependent step against round 3's hybrid (26.3 ms per 16 384 rays)
set -u
mkdir -p gpurun_out
python -m pytest tests/test_training.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -n 25 > gpurun_out/r04_c10_train_tests.log
tail -n 25 gpurun_out/r04_c10_train_tests.log
for n in 1024 16384; do
  python tools/train_step_profile.py $n bf16 --views 2>&1 | grep -v amdgpu.ids | tail -n 1
done | tee gpurun_out/r04_c10_views_step.txt
python tools/train_step_profile.py 16384 f32 --views 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a gpurun_out/r04_c10_views_step.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_v
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o s -- python $GRAFT_REPO_ROOT/tools/train_step_profile.py 16384 bf16 --views > /tmp/prof_v.log 2>&1
db=$(find /tmp/prof_v -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py "$db" 2>&1 | head -24 | cut -c1-200 > $GRAFT_REPO_ROOT/gpurun_out/r04_train_kernel_stats_16384_views.txt
head -16 $GRAFT_REPO_ROOT/gpurun_out/r04_train_kernel_stats_16384_views.txt
