# round 5, GPU session 9: the fused training loss (nrnerf_loss_*) -- parity, the step's launch count and time; then the whole GPU tier
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c9; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_training.py -x -q -k "fused_loss or with_the_fused_loss" > gpurun_out/c9/pytest_loss.txt 2>&1; tail -5 gpurun_out/c9/pytest_loss.txt
timeout 600 python tools/train_step_profile.py 1024 bf16 2>&1 | tail -3
timeout 600 python tools/experiments/step_kernel_sequence.py > gpurun_out/c9/train_step_kernel_sequence_1024.txt 2>&1; head -3 gpurun_out/c9/train_step_kernel_sequence_1024.txt; tail -2 gpurun_out/c9/train_step_kernel_sequence_1024.txt
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/c9/pytest_gpu_full.txt 2>&1; tail -8 gpurun_out/c9/pytest_gpu_full.txt
