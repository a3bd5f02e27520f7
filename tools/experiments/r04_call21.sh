#!/bin/bash
# round 4, GPU call 21 (tools/with_reference.sh): the whole GPU tier on the final tree with the reference staged, smoke, bench
set -u
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -n 6 | tee gpurun_out/r04_gpu_suite.txt
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -n 7 | tee gpurun_out/r04_smoke.txt
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
r=d['roofline']; t=d['train_step']
print(d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), t['ms_per_step'], t['hip_graph']['ms_per_step'])"
