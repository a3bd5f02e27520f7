# round 5, GPU session 2: fit the non-compiled acceptance checkpoint (coarse 192 / fine 320 wide); the new >256-sample gradient and
# 16-bit render tests; the fitted-checkpoint acceptance tests incl. the generic-kernel routes; a bench line with the new fields
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c2; export TMPDIR=/tmp
python oracle/fit_checkpoint.py --arch w192_320 --minutes 4.0 --out gpurun_out/fitted_w192_320.tar > gpurun_out/c2/fit_w192_320.log 2>&1 &
FIT=$!
timeout 1200 python -m pytest tests/test_training.py -x -q -k "fp32_gradients or bf16_gradients or native_bender_forward" > gpurun_out/c2/pytest_training.txt 2>&1; tail -4 gpurun_out/c2/pytest_training.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "large_sample_counts" > gpurun_out/c2/pytest_large.txt 2>&1; tail -4 gpurun_out/c2/pytest_large.txt
wait $FIT; tail -3 gpurun_out/c2/fit_w192_320.log
cp gpurun_out/fitted_w192_320.tar tests/golden/
timeout 1200 python -m pytest tests/test_fitted_checkpoint.py -x -q -s > gpurun_out/c2/pytest_fitted.txt 2>&1; tail -4 gpurun_out/c2/pytest_fitted.txt
python bench.py --no-train-step > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err; tail -c 600 gpurun_out/c2/bench.json
