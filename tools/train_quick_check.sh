#!/bin/bash
# GPU box: the quick loop used while tuning the training kernels -- training / distributed GPU tests, the per-kernel averages of
# a 16384-ray shipped-recipe step (rocprofv3 --kernel-trace --stats) and the step times of the three modes at three batch sizes.
R=$PWD
timeout 900 python -m pytest tests/test_training.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tq && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tq -o s -- python $R/tools/train_step_profile.py 16384 bf16 > /tmp/tq.log 2>&1
db=$(find /tmp/tq -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$db" 2>/dev/null | grep -E "trunk_|bend_|composite|operands" | head -14
cd $R
python - <<'PY'
import torch
from nonrigid_nerf_amd import training
from nonrigid_nerf_amd.synthetic import SceneConfig
for n in (1024, 4096, 16384):
    r = training.bench_train_step(None, SceneConfig(), torch.device("cuda:0"), precision="bf16", n_rays=n, steps=10 if n > 4096 else 30, warmup=3)
    print(n, "shipped", round(r["ms_per_step"], 3), "graph", r["hip_graph"].get("ms_per_step"), "data_only", r["data_term_only"]["ms_per_step"])
PY
