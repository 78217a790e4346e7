#!/usr/bin/env python
"""Vendor-library yardstick for the ViT GEMM shapes on this box (torch.matmul -> hipBLASLt / rocBLAS), same shapes, data
distribution and steady-state protocol (15 warm-up + 40 timed launches per shape) as tools/gemm_bench.py.  NOT part of the
product path: the product's GEMM is csrc/gemm_f16_p8.hip; this only answers "what does the vendor's best plain fp16 GEMM do
on K = 1280 shapes" (VERDICT r4).  The vendor call is a PLAIN product with an fp16 result — no bias, GELU, fp32 output or
in-place residual — so it is an upper yardstick for the K-loop, not a like-for-like replacement of a fused launch.

    python tools/blas_ceiling.py [frames=8]"""
import sys

import torch

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D = 1280
Ml, Mg = B * 2688, B * 4096
shapes = [("qkv  live", Ml, 3 * D, D), ("proj live", Ml, D, D), ("fc1  live", Ml, 4 * D, D), ("fc2  live", Ml, D, 4 * D),
          ("qkv  glob", Mg, 3 * D, D), ("proj glob", Mg, D, D), ("fc1  glob", Mg, 4 * D, D), ("fc2  glob", Mg, D, 4 * D),
          ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]
g = torch.Generator().manual_seed(0)
for zero in (False, True):
    tot_f = tot_t = 0.0
    for (name, M, N, K) in shapes:
        A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        if zero:
            A.zero_(), W.zero_()
        C = torch.empty(M, N, dtype=torch.float16, device=dev)
        for _ in range(15):
            torch.matmul(A, W.t(), out=C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        e0.record()
        for _ in range(reps):
            torch.matmul(A, W.t(), out=C)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        fl = 2.0 * M * N * K
        print(f"{'zeros ' if zero else 'random'} {name:12s} M={M:6d} N={N:5d} K={K:5d} {t * 1e6:9.1f} us {fl / t / 1e12:7.1f} TFLOP/s", flush=True)
        if not name.startswith("square"):
            cnt = 7 if "live" in name else 25                       # ViT-H: 7 blocks on the live rows, 25 on the full grid
            tot_f += fl * cnt
            tot_t += t * cnt
        del A, W, C
    print(f"{'zeros ' if zero else 'random'} ViT-H block mix (7 live + 25 full blocks): {tot_f / tot_t / 1e12:7.1f} TFLOP/s = {tot_f / tot_t / 2.5e15:.3f} of the dense fp16 peak")
