#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -q -m gpu -k "full_size or rccl or split_bender_path_at or bench_spawns" > gpurun_out/c10_tests.log 2>&1; tail -3 gpurun_out/c10_tests.log
timeout 300 python tools/render_path_bench.py 120 2>&1 | tail -2
timeout 300 python tools/render_path_bench.py 16 1080 1920 2>&1 | tail -2
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"; }
timeout 200 $B --rays 1024 --steps 300 2>&1 | tail -1 | show "1024 rays"
timeout 200 $B --rays 32768 --steps 60 2>&1 | tail -1 | show "32768 rays"
timeout 200 $B --rays 65536 --steps 40 --precision f16 2>&1 | tail -1 | show "config5 chunk 65536 f16"
timeout 200 $B --rays 2073600 --steps 3 --warmup 1 2>&1 | tail -1 | show "1080p frame bf16"
timeout 200 $B --use-viewdirs 2>&1 | tail -1 | show "viewdirs (synthetic)"
timeout 200 $B --use-viewdirs --bend-depth 7 2>&1 | tail -1 | show "config4 viewdirs deep bender"
timeout 200 $B --use-viewdirs --bend-depth 7 --exact-viewdirs 2>&1 | tail -1 | show "config4 exact viewdirs"
timeout 300 $B --precision f32 --steps 3 --warmup 1 2>&1 | tail -1 | show "f32"
