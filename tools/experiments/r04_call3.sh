#!/bin/bash
# round 4, GPU call 3: fused compositing with the inputs prefetched one tile ahead: bit-identity again, A/B bench twice (alternating)
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fused_into" > gpurun_out/r04_c3_fused.log 2>&1
echo "fused rc=$?"; tail -n 4 gpurun_out/r04_c3_fused.log
for rep in 1 2; do for U in 0 1; do
  NRNERF_UNFUSED_COMPOSITE=$U python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 > gpurun_out/r04_c3_bench_u${U}_$rep.json 2> gpurun_out/r04_c3_bench_u${U}_$rep.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c3_bench_u${U}_$rep.json").read().strip().splitlines()[-1])
print("rep $rep unfused=$U", d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"], d["roofline"]["frac"])
PY
done; done
python bench.py --steps 20 --warmup 5 --rays 1024 --no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1024 rays', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
