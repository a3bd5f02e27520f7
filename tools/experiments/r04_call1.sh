#!/bin/bash
# round 4, GPU call 1: fit the two missing architecture families with the oracle (in parallel: eager torch leaves the GPU
# mostly idle), then the accuracy bar on all three fitted checkpoints, the pinned fp32 numbers, the two-pass backward.
set -u
mkdir -p gpurun_out
python oracle/fit_checkpoint.py --arch config4 --iters 6000 --minutes 4.5 --out gpurun_out/fitted_config4.tar > gpurun_out/r04_fit_config4.log 2>&1 &
P1=$!
python oracle/fit_checkpoint.py --arch w128 --iters 6000 --minutes 4.5 --out gpurun_out/fitted_w128.tar > gpurun_out/r04_fit_w128.log 2>&1 &
P2=$!
wait $P1 $P2
tail -3 gpurun_out/r04_fit_config4.log gpurun_out/r04_fit_w128.log
cp gpurun_out/fitted_config4.tar gpurun_out/fitted_w128.tar tests/golden/
rm -f gpurun_out/r04_pins.jsonl
timeout 900 python -m pytest tests/test_fitted_checkpoint.py -q -s -m gpu > gpurun_out/r04_fitted_tests.log 2>&1
echo "fitted tests rc=$?"; grep -a "fitted checkpoint\|passed\|failed\|Error" gpurun_out/r04_fitted_tests.log | tail -20
NRNERF_PIN_RECORD=gpurun_out/r04_pins.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "golden or 4k_rays or full_frame or every_compiled_variant" > gpurun_out/r04_parity_pins.log 2>&1
echo "parity rc=$?"; grep -a "pinned fp32\|passed\|failed\|Error\|196 608" gpurun_out/r04_parity_pins.log | tail -30
timeout 600 python -m pytest tests/test_training.py -q -s -m gpu -k "two_pass" > gpurun_out/r04_two_pass.log 2>&1
echo "two-pass rc=$?"; grep -a "two-pass\|passed\|failed\|Error" gpurun_out/r04_two_pass.log | tail -10
