#!/bin/bash
# round 4, GPU call 13: the driver's round-end sequence on the current tree -- GPU tests, smoke, default bench
set -u
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -n 8 | tee gpurun_out/r04_c13_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 6 | tee gpurun_out/r04_c13_smoke.log
python bench.py > gpurun_out/r04_c13_bench.json 2> gpurun_out/r04_c13_bench.err
echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r04_c13_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms_per_step'], d.get('train_step'))"
