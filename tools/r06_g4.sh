set -x
python -m pytest tests/test_training.py -q -x -k "tn_products or encoding_rows or fused_adam or generic" 2>&1 | tail -30 > gpurun_out/r06_gen_tests.txt
python -m pytest tests/test_training.py -q -s -k "trains_faster_than_eager" 2>&1 | grep -E "ms|passed|failed" > gpurun_out/r06_gen_speed.txt
python tools/train_step_sequence.py 1024 bf16 2>&1 | grep -E "adam|busy|^#" > gpurun_out/r06_adam_time.txt
