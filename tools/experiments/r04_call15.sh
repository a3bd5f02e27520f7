#!/bin/bash
# round 4, GPU call 15: the packed weights follow a FUSED optimiser's step (render._watch_optimizers); what the eager training step costs
# now that its weight refresh really runs (until now the harness' fused Adam left the version counters alone: no refresh, stale weights)
set -u
mkdir -p gpurun_out
python -m pytest tests/test_training.py -q -m gpu -k "optimiser_step_of_any_kind" 2>&1 | grep -v amdgpu.ids | tail -n 8
python tools/experiments/debug_refresh_path.py 2>&1 | grep -v amdgpu.ids | tail -n 3
python tools/train_step_scaling.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_train_step_modes.txt
for n in 1024 16384; do python tools/train_step_profile.py $n bf16 --views 2>&1 | grep -v amdgpu.ids | tail -n 1; done | tee gpurun_out/r04_views_train_step.txt
python tools/train_step_profile.py 16384 f32 --views 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a gpurun_out/r04_views_train_step.txt
