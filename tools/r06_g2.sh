set -x
./tools/probes/tr_probe.bin > gpurun_out/r06_tr_b16_probe.txt 2>&1
python -m pytest tests/test_training.py -q -x -s -k "fused_adam or fused_optimiser or hip_graph or optimiser_step_of_any_kind" 2>&1 | tail -25 > gpurun_out/r06_adam_tests.txt
python -m pytest tests/test_gpu_parity.py -q -s -k "f16 or full_chunk" 2>&1 | tail -25 > gpurun_out/r06_f16_tests.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --min-gpu-seconds 0 2>&1 | tail -1 > gpurun_out/r06_bench_a.json
python bench.py --steps 4 --warmup 2 --precision f16 --rays 2073600 --chunk 65536 --max-rays-per-launch 65536 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 > gpurun_out/r06_config5_a.json
