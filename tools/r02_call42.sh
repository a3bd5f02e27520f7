#!/bin/bash
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --min-gpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['train_step']['ms_per_step'])"
