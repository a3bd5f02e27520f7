set -x
python -m pytest tests/test_gpu_parity.py -q -x -k "coarse_epilogue or width_class_kernel_with_fused" 2>&1 | tail -15 > gpurun_out/r06_g1_tests.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
for i in 1 2 3; do
  for e in 0 1; do
    echo "epilogue=$e run $i" >> gpurun_out/r06_coarse_epilogue_ab.txt
    NRNERF_FUSED_COARSE_EPILOGUE=$e $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernels_ms_per_step'], r['kernels_launched'])" >> gpurun_out/r06_coarse_epilogue_ab.txt 2>&1
  done
done
for u in 0 1; do
  echo "w192 unfused_composite=$u" >> gpurun_out/r06_gx16_fused_ab.txt
  NRNERF_UNFUSED_COMPOSITE=$u $B --netwidth 192 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernels_ms_per_step'], r['kernels_launched'])" >> gpurun_out/r06_gx16_fused_ab.txt 2>&1
done
python tools/power_trace.py bf16 250 > gpurun_out/r06_power_trace.txt 2>&1
python bench.py --frames 8 --steps 3 --warmup 1 2>&1 | tail -1 > gpurun_out/r06_frames8.json
