set -x
python -m pytest tests/test_training.py -q -x -s -k "fused_adam or fused_optimiser" 2>&1 | tail -12 > gpurun_out/r06_adam_tests.txt
python tools/train_step_sequence.py 1024 bf16 > gpurun_out/r06_train_step_kernel_sequence_1024.txt 2>&1
python tools/train_step_sequence.py 1024 bf16 --torch-adam > gpurun_out/r06_train_step_kernel_sequence_1024_torch_adam.txt 2>&1
python -m pytest tests/test_fitted_checkpoint.py -q -s -m gpu 2>&1 | grep -E "fitted checkpoint|passed|failed" > gpurun_out/r06_fitted_accuracy.txt
