#!/usr/bin/env python
"""Micro-benchmark of the fp16 ViT GEMM shapes through the C ABI (HIP events on the launching stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ZERO = len(sys.argv) > 2 and sys.argv[2] == "zeros"     # zero-filled operands: separates power (DVFS) limits from stalls
D = 1280
shapes = [(B * 4900, 3 * D, D, 2), (B * 4096, 3 * D, D, 2), (B * 4900, D, D, 1), (B * 4096, D, D, 1), (B * 4096, 4 * D, D, 2),
          (B * 4096, D, 4 * D, 1), (B * 4096, D, 768, 1), (B * 4096, 256, D, 1), (4096, 4096, 4096, 1), (8192, 8192, 8192, 1)]
g = torch.Generator().manual_seed(0)
for (M, N, K, dt) in shapes:
    A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
    if ZERO:
        A.zero_(), W.zero_()
    bias = torch.zeros(N, device=dev)
    Cc = torch.empty(M, N, device=dev, dtype=torch.float16 if dt == 2 else torch.float32)
    for _ in range(3):
        _lib.check(lib.sampt_gemm(dt, _lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(Cc), M, N, K, 0, 1.0, _lib.stream_ptr()), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        lib.sampt_gemm(dt, _lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(Cc), M, N, K, 0, 1.0, _lib.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print(f"M={M:6d} N={N:5d} K={K:5d} out={'f16' if dt == 2 else 'f32'}  {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s")
    del A, W, Cc
