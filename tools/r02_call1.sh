#!/bin/bash
# round-2 GPU call 1: fit the checkpoint, accuracy tests on it, full GPU suite, bench on both scenes
mkdir -p gpurun_out
python oracle/fit_checkpoint.py --iters 12000 --minutes 5 --out gpurun_out/fitted_latest.tar > gpurun_out/fit.log 2>&1
tail -5 gpurun_out/fit.log
cp gpurun_out/fitted_latest.tar tests/golden/fitted_latest.tar
timeout 600 python -m pytest tests/test_fitted_checkpoint.py -q -s -m gpu > gpurun_out/fitted_tests.log 2>&1
tail -15 gpurun_out/fitted_tests.log
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_fitted_checkpoint.py > gpurun_out/gpu_tests.log 2>&1
tail -5 gpurun_out/gpu_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_synth.log 2>&1
tail -1 gpurun_out/bench_synth.log | cut -c1-1500
timeout 300 python bench.py --steps 20 --warmup 5 --scene fitted --no-cpu-baseline > gpurun_out/bench_fitted.log 2>&1
tail -1 gpurun_out/bench_fitted.log | cut -c1-1500
