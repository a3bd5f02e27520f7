#!/bin/bash
# round-2 GPU call 2: split-bender path -- bit-identity + golden tests, A/B against the fused fine pass, kernel stats
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "split_bender or golden or tiny_and_ragged or surface or stochastic or full_size" > gpurun_out/c2_split_tests.log 2>&1
tail -4 gpurun_out/c2_split_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
for i in 1 2; do
NRNERF_FUSED_FINE_BENDER=1 timeout 200 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fused', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"
timeout 200 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('split', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"
done
timeout 200 $B --scene synthetic 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('split synthetic', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"
timeout 200 $B --precision f16 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('split f16', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"
timeout 200 $B --rays 1024 --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('split 1024 rays', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"
timeout 200 $B --rays 32768 --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('split 32768 rays', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"
R=$PWD; OUT=$R/gpurun_out/prof_c2; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-psnr --min-gpu-seconds 0 > $OUT/stats.log 2>&1)
DB=$(find $OUT/stats -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" > gpurun_out/c2_kernel_stats.txt 2>&1
find $OUT -name "*.db" -delete
head -12 gpurun_out/c2_kernel_stats.txt | cut -c1-200
