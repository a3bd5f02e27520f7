set -x
python tools/train_step_sources.py 1024 bf16 > gpurun_out/r06_train_step_sources.txt 2>&1
tail -100 gpurun_out/r06_train_step_sources.txt
