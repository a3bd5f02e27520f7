python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3 > gpurun_out/r06_train_suite.txt
python tools/generic_step_sequence.py 192 2>&1 | grep -v Warning > gpurun_out/r06_generic_w192_step_sequence.txt
python tools/generic_step_sequence.py 256 2>&1 | grep -v Warning | tail -1 > gpurun_out/r06_generic_w256_busy.txt
python tools/generic_step_sequence.py 192 --bender 2>&1 | grep -v Warning | tail -1 >> gpurun_out/r06_generic_w256_busy.txt
python -m pytest tests/test_training.py -q -s -k "trains_faster_than_eager" 2>&1 | grep -E "ms|passed|failed" > gpurun_out/r06_gen_speed.txt
