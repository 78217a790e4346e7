#!/usr/bin/env python
"""The PIPS mixer's channel-MLP GEMMs as the tracker runs them: fc1 (K = 512 -> 2048, GELU) and fc2 (2048 -> 512, + residual)
back to back over 12 different weight sets (so weights stream from the Infinity Cache, not from a hot L2), M = 8 x chains rows.
python tools/thin_bench.py [chains ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P, S = _lib.ptr, _lib.stream_ptr
g = torch.Generator().manual_seed(0)
for chains in [int(a) for a in sys.argv[1:]] or [8, 16, 48]:
    M = 8 * chains
    w1 = [(torch.randn(2048, 512, generator=g) / 22).to(dev) for _ in range(12)]
    w2 = [(torch.randn(512, 2048, generator=g) / 45).to(dev) for _ in range(12)]
    b1, b2 = torch.zeros(2048, device=dev), torch.zeros(512, device=dev)
    x = torch.randn(M, 512, generator=g).to(dev)
    h = torch.empty(M, 2048, device=dev)
    y = torch.empty(M, 512, device=dev)

    def chain():
        for i in range(12):
            lib.sampt_gemm(0, P(x), P(w1[i]), P(b1), None, P(h), M, 2048, 512, 2, 1.0, S())
            lib.sampt_gemm(0, P(h), P(w2[i]), P(b2), P(x), P(y), M, 512, 2048, 0, 1.0, S())
    for _ in range(3):
        chain()
    ref = torch.nn.functional.gelu(x.double() @ w1[11].double().t()) @ w2[11].double().t() + x.double()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        chain()
    e1.record()
    torch.cuda.synchronize()
    print(f"M={M:4d}: fc1 + fc2 pair {e0.elapsed_time(e1) / 120 * 1e3:7.1f} us (rel err {err:.1e})", flush=True)
