#!/usr/bin/env python
"""TEMPORARY: where the non-K-loop time of gemm_f16_p8 goes (SAMPT_P8_DBG bits: 1 no stores, 2 no residual loads, 4 no epilogue,
8 no GELU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P, S = _lib.ptr, _lib.stream_ptr
g = torch.Generator().manual_seed(0)
D, Mg = 1280, 8 * 4096
shapes = [("qkv", Mg, 3 * D, D, 2, 0, False), ("proj", Mg, D, D, 1, 0, True), ("fc1", Mg, 4 * D, D, 2, 2, False),
          ("fc2", Mg, D, 4 * D, 1, 0, True), ("fc1-as-qkvN", Mg, 3 * D, D, 2, 2, False), ("qkv-as-fc1N", Mg, 4 * D, D, 2, 0, False),
          ("sq4096", 4096, 4096, 4096, 2, 0, False)]
for (name, M, N, K, dt, act, use_res) in shapes:
    A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
    bias = torch.zeros(N, device=dev)
    Cc = torch.zeros(M, N, device=dev, dtype=torch.float16 if dt == 2 else torch.float32)
    res = Cc if use_res else None
    line = f"{name:12s} M={M} N={N} K={K}:"
    for dbg in (0, 1, 2, 3, 4, 8):
        if (dbg & 2) and not use_res or (dbg & 8) and act != 2:
            continue
        os.environ["SAMPT_P8_DBG"] = str(dbg)
        call = lambda: lib.sampt_gemm_ex(dt, P(A), P(W), P(bias), P(res), P(Cc), M, N, K, act, 1.0, None, None, 0, 0, S())
        for _ in range(3):
            _lib.check(call(), "gemm")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        line += f"  dbg{dbg}: {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:6.0f} TF"
    os.environ["SAMPT_P8_DBG"] = "0"
    print(line, flush=True)
    del A, W, Cc
