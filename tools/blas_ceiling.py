#!/usr/bin/env python
"""Vendor-library ceiling for the ViT GEMM shapes on this box (torch.matmul -> hipBLASLt/rocBLAS), same data
distribution as tools/gemm_bench.py.  A yardstick for our hand-written kernel, not part of the product path."""
import sys

import torch

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D = 1280
shapes = [(B * 4900, 3 * D, D), (B * 4096, 3 * D, D), (B * 4900, D, D), (B * 4096, 4 * D, D), (B * 4096, D, 4 * D),
          (4096, 4096, 4096), (8192, 8192, 8192)]
g = torch.Generator().manual_seed(0)
for zero in (False, True):
    for (M, N, K) in shapes:
        A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        if zero:
            A.zero_(), W.zero_()
        for _ in range(3):
            C = A @ W.t()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            C = A @ W.t()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"{'zeros ' if zero else 'random'} M={M:6d} N={N:5d} K={K:5d} {t * 1e6:9.1f} us {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s")
