#!/bin/bash
# round 4, GPU call 40 (tools/with_reference.sh): final state with the 16x16x32 trunk-only kernel -- rendering trace + PMC passes on the
# final device sources (hash guard of roofline.traffic), the whole GPU tier with the reference staged, smoke, the default bench line
set -u
mkdir -p gpurun_out
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1; echo "collect rc=$?"
python -m pytest tests/ -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -n 4 | tee gpurun_out/r04_gpu_suite.txt
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | grep "smoke\]" | tee gpurun_out/r04_smoke.txt
env -u NRNERF_REFERENCE python bench.py > gpurun_out/r04_bench_bf16.json 2> gpurun_out/r04_bench_bf16.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_bf16.json').read().strip().splitlines()[-1])
r=d['roofline']; t=d['train_step']
print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('frac_issued_mfma'), r.get('traffic'), r['kernels_ms_per_step'], r['coarse_pass']['frac'], r['library_gemm_tflops_same_box'])
print('train', t['ms_per_step'], t['final_loss'], t['hip_graph']['ms_per_step'], t['roofline']['frac'], t['roofline']['mfma']['frac'])
print(d['cpu_baseline']['value'], d['cpu_baseline'].get('thread_sweep_rays_per_s'), d['psnr_vs_oracle_db']['rgb_map'])"
