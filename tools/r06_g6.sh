python -m pytest tests/test_training.py tests/test_c_abi.py tests/test_fitted_checkpoint.py -q -x -m gpu 2>&1 | tail -6 > gpurun_out/r06_train_suite.txt
python tools/train_step_sequence.py 1024 bf16 > gpurun_out/r06_train_step_kernel_sequence_1024.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --min-gpu-seconds 0 2>&1 | tail -1 > gpurun_out/r06_bench_b.json
python -m pytest tests/test_training.py -q -s -k "trains_faster_than_eager" 2>&1 | grep -E "ms|passed|failed" > gpurun_out/r06_gen_speed.txt
