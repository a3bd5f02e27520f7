set -x
python -m pytest tests/test_training.py tests/test_install_reference.py -m gpu -q > gpurun_out/r06_p1_tests.txt 2>&1; tail -5 gpurun_out/r06_p1_tests.txt
python tools/train_step_sequence.py 1024 bf16 > gpurun_out/r06_p1_sequence.txt 2>&1; grep "^#" gpurun_out/r06_p1_sequence.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --min-gpu-seconds 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d["train_step"], indent=None)[:1500])'
