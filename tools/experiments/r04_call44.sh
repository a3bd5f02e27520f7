#!/bin/bash
# round 4, GPU call 44: sanity of the library as rebuilt from the committed sources: smoke, the quick bench line, two parity tests
set -u
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | grep "smoke\]"
python bench.py --no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['traffic'], r['kernels_ms_per_step'])"
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or headline" 2>&1 | grep -v amdgpu.ids | tail -n 2
