#!/bin/bash
# round-2 GPU call 3: training path tests, split-bender (both passes) tests, A/B of the bender variants
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -q -x -m gpu -s > gpurun_out/c3_training_tests.log 2>&1
tail -25 gpurun_out/c3_training_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "split_bender or golden or tiny_and_ragged or surface or stochastic or full_size or boundary_contract or changed_weights" > gpurun_out/c3_split_tests.log 2>&1
tail -6 gpurun_out/c3_split_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"; }
for i in 1 2; do
NRNERF_SPLIT_COARSE=0 timeout 200 $B 2>&1 | tail -1 | show "split fine only"
timeout 200 $B 2>&1 | tail -1 | show "split both    "
done
NRNERF_FUSED_FINE_BENDER=1 timeout 200 $B 2>&1 | tail -1 | show "fused         "
timeout 200 $B --rays 1024 --steps 200 2>&1 | tail -1 | show "split both 1024 rays"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-psnr --min-gpu-seconds 0 --train-step 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train_step', d.get('train_step'))"
