#!/bin/bash
# round 4, GPU call 20 (tools/with_reference.sh): the reference's REAL modules, training iteration with exact Jacobian view directions
# after install(); step time of the native exact-direction iteration
set -u
mkdir -p gpurun_out
python -m pytest tests/test_install_reference.py -q -m gpu -s -k "training_iteration" 2>&1 | grep -v amdgpu.ids | grep "^\[\|passed\|failed\|Error\|assert" | tee gpurun_out/r04_install_reference_exact.txt
for n in 1024 16384; do python tools/train_step_profile.py $n bf16 --exact 2>&1 | grep -v amdgpu.ids | tail -n 1; done | tee gpurun_out/r04_exact_train_step.txt
