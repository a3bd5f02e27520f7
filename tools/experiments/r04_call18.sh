#!/bin/bash
# round 4, GPU call 18: the committed evidence re-collected on the final device sources -- rendering trace + PMC passes (source-hash
# guard of roofline.traffic), training trace + HBM-traffic PMC passes at 1024 / 16384 rays, the default bench line
set -u
mkdir -p gpurun_out
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
echo "collect rc=$?"
bash tools/collect_train_profiles.sh r04 > gpurun_out/r04_collect_train.log 2>&1
echo "collect_train rc=$?"
python bench.py > gpurun_out/r04_bench_bf16.json 2> gpurun_out/r04_bench_bf16.err
echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_bf16.json').read().strip().splitlines()[-1])
r=d['roofline']; t=d['train_step']
print(d['value'], d['ms_per_step'], r['frac'], r.get('frac_issued_mfma'), r.get('traffic'), r['kernels_ms_per_step'])
print('train', t['ms_per_step'], t['final_loss'], t['hip_graph']['ms_per_step'], t['hip_graph']['final_loss'], t['roofline']['frac'], t['roofline']['mfma']['frac'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['thread_sweep_rays_per_s'])"
