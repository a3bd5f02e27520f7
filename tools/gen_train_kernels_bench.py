#!/usr/bin/env python
"""GPU box: nrnerf_tn_products / nrnerf_adam_step alone (HIP-event timed): the weight-gradient products of a D x W trunk over M samples, bf16 and
fp32, against torch.bmm's route of round 5; the fused optimiser step against torch.optim.Adam(fused).
    python tools/gen_train_kernels_bench.py [W] [D] [rays] [samples]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nonrigid_nerf_amd import training  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
S = int(sys.argv[4]) if len(sys.argv) > 4 else 192
dev = torch.device("cuda:0")
M = N * S


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for dt in (torch.bfloat16, torch.float32):
    acts = torch.randn(D, M, W, device=dev).to(dt)
    d_pre = torch.randn(D, M, W, device=dev).to(dt)
    enc = torch.randn(M, 64, device=dev).to(dt)
    jobs, off = [], 0
    for i in range(D):
        if i == 0:
            jobs.append((d_pre[0], 0, enc, 0, W, 63, 63, off, off + W * 63)); off += W * 63 + W
        else:
            jobs.append((d_pre[i], 0, acts[i - 1], 0, W, W, W, off, off + W * W)); off += W * W + W
    ms = timed(lambda: training._tn_products(jobs, M, off, dev))
    flops = 2.0 * M * (W * 63 + (D - 1) * W * W)
    gbytes = (2 * D * M * W + M * 64) * acts.element_size() / 1e9
    def bmm():
        for i in range(1, D):
            a3 = d_pre[i].view(M // 2048, 2048, W).transpose(1, 2)
            (torch.bmm(a3, acts[i - 1].view(M // 2048, 2048, W)) if dt == torch.float32 else torch.bmm(a3, acts[i - 1].view(M // 2048, 2048, W), out_dtype=torch.float32)).sum(0)
            d_pre[i].sum(0, dtype=torch.float32)
    try:
        ms_b = timed(bmm)
    except Exception as e:
        ms_b = float("nan")
    print(f"[tn_products {str(dt).split('.')[-1]}] W {W} D {D} M {M}: {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s, arrays {gbytes:.2f} GB -> {gbytes / ms * 1e3:.0f} GB/s "
          f"(torch.bmm chunks + column sums: {ms_b:.3f} ms)")
    del acts, d_pre

# the optimiser step: 1.2 M parameters in ~50 tensors
ps = [torch.nn.Parameter(torch.randn(256, 256, device=dev)) for _ in range(16)] + [torch.nn.Parameter(torch.randn(256, device=dev)) for _ in range(34)]
for p in ps:
    p.grad = torch.randn_like(p)
fa = training.FusedAdam(ps, lr=1e-3)
ta = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3, fused=True)
for p, q in zip(ta.param_groups[0]["params"], ps):
    p.grad = q.grad.clone()
print(f"[adam, 50 separate tensors] FusedAdam {timed(fa.step, 20) * 1e3:.1f} us ({fa.last_segments} segments), torch fused {timed(ta.step, 20) * 1e3:.1f} us")
flat_g = torch.randn(sum(p.numel() for p in ps), device=dev)
o = 0
net = torch.nn.Sequential(*[torch.nn.Linear(256, 256) for _ in range(16)]).to(dev)
