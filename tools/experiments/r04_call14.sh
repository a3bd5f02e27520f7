#!/bin/bash
# round 4, GPU call 14 (through tools/with_reference.sh): install() on the reference's REAL modules on the final training kernels,
# now also with use_viewdirs; and the time-conditioned + view-dependent gradient case
set -u
mkdir -p gpurun_out
python -m pytest tests/test_install_reference.py -q -m gpu -s 2>&1 | grep -v amdgpu.ids | grep "^\[\|passed\|failed\|Error\|assert" | tee gpurun_out/r04_install_reference_gpu.txt
python -m pytest tests/test_training.py -q -m gpu -k "time_conditioned_viewdirs" 2>&1 | grep -v amdgpu.ids | tail -n 12
