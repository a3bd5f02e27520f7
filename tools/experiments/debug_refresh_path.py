#!/usr/bin/env python
"""GPU box: does an optimiser step make the next get_model refresh the packed weights?  (fused / foreach / plain Adam)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import render as R, training  # noqa: E402
from nonrigid_nerf_amd.synthetic import SceneConfig  # noqa: E402

dev = torch.device("cuda:0")
calls = {"device": 0, "host": 0}
ud, uh = R.Model.update_from_device, R.Model.update
R.Model.update_from_device = lambda self, *a, **k: (calls.__setitem__("device", calls["device"] + 1), ud(self, *a, **k))[1]
R.Model.update = lambda self, *a, **k: (calls.__setitem__("host", calls["host"] + 1), uh(self, *a, **k))[1]
for kind in ("fused", "foreach", "plain"):
    rb, coarse, fine = training._fresh_training_modules(SceneConfig(), dev, 64)
    params = [p for m in (rb, coarse, fine) for p in m.parameters()]
    for p in params:
        p.requires_grad_(True)
    opt = torch.optim.Adam(params, lr=1e-3, fused=(kind == "fused"), foreach=(kind == "foreach") if kind != "fused" else None)
    R.set_precision("bf16")
    m0 = R.get_model(coarse, fine, precision="bf16", device=dev)
    v0 = params[0]._version
    for p in params:
        p.grad = torch.ones_like(p)
    opt.step()
    v1 = params[0]._version
    calls["device"] = calls["host"] = 0
    m1 = R.get_model(coarse, fine, precision="bf16", device=dev)
    print(f"{kind:8s}: _version {v0} -> {v1}; get_model after the step: same handle {m1 is m0}, device refreshes {calls['device']}, host refreshes {calls['host']}")
