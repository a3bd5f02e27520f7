set -x
bash tools/collect_config_evidence.sh r06 > gpurun_out/r06_config_evidence.log 2>&1
tail -12 gpurun_out/r06_config_evidence.log
bash tools/collect_train_profiles.sh r06 > gpurun_out/r06_train_profiles.log 2>&1
python bench.py --frames 8 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r06_frames8.json
python tools/generic_step_sequence.py > gpurun_out/r06_generic_w192_step_sequence.txt 2>&1
python -m pytest tests/test_training.py -m gpu -q -s -k "trains_faster_than_eager" 2>&1 | grep "generic training" > gpurun_out/r06_gen_speed.txt
python tools/train_step_sequence.py 1024 bf16 --torch-adam 2>&1 | grep "^#" > gpurun_out/r06_train_step_torch_adam_summary.txt
cat gpurun_out/r06_gen_speed.txt gpurun_out/r06_train_step_torch_adam_summary.txt
