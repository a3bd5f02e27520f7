#!/usr/bin/env python
"""What the matrix pipe sustains on THIS box for a plain library GEMM (hipBLASLt through torch), as context for the
fused kernel's roofline fraction: the 2.5 PFLOP/s dense bf16 peak assumes 2.4 GHz, the chip clocks down under MFMA load.
  python tools/gemm_ceiling.py        -> TFLOP/s for square bf16 GEMMs and for the trunk's [M,256]x[256,256] shape."""
import time

import torch


def bench(m, n, k, dtype=torch.bfloat16, iters=30):
    a = torch.randn(m, k, device="cuda", dtype=dtype)
    b = torch.randn(k, n, device="cuda", dtype=dtype)
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / iters
    return 2.0 * m * n * k / dt / 1e12, dt * 1e3


if __name__ == "__main__":
    for shape in [(8192, 8192, 8192), (16384, 16384, 8192), (196608 * 32, 256, 256), (196608 * 8, 256, 256)]:
        tf, ms = bench(*shape)
        print(f"bf16 GEMM M={shape[0]} N={shape[1]} K={shape[2]}: {tf:8.1f} TFLOP/s  ({ms:.3f} ms)", flush=True)
