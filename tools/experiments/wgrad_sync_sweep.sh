#!/bin/bash
# GPU box: trunk_wgrad re-alignment barrier sweep (NRNERF_WGRAD_SYNC = pairs of blocks between workgroup barriers, 0 = never):
# whole-step time of the shipped recipe at 16384 and 1024 rays, and trunk_wgrad's own average from the profiler table.
for s in 0 2 8 32 128 512; do
    echo "== NRNERF_WGRAD_SYNC=$s"
    NRNERF_WGRAD_SYNC=$s python tools/train_step_profile.py 16384 bf16 2>&1 | tail -1
    NRNERF_WGRAD_SYNC=$s python tools/train_step_profile.py 16384 bf16 --table 2>&1 | grep -E "trunk_wgrad" | head -1 | awk '{print "   trunk_wgrad avg:", $(NF-1), "calls", $NF}'
    NRNERF_WGRAD_SYNC=$s python tools/train_step_profile.py 1024 bf16 2>&1 | tail -1
done
