#!/usr/bin/env python
"""Micro-benchmark of the fused ViT attention kernels through the C ABI (ViT-H geometry, 8 frames): the fp16 kernel and, with
``x3`` on the command line, the split-fp16 one (TFLOP/s fp32-equivalent: each product costs 3 MFMAs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd import _lib  # noqa: E402

if os.environ.get("ATTN_BENCH_LIB"):                     # A/B against another build of the library (tools/README.md)
    _lib.LIB_PATH = os.path.abspath(os.environ["ATTN_BENCH_LIB"])
lib = _lib.load()
dev = torch.device("cuda:0")
heads, hd = 16, 80
D = heads * hd
X3 = "x3" in sys.argv
g = torch.Generator().manual_seed(0)
for name, B, S_ in (("window", 200, 14), ("global", 8, 64)):
    N = S_ * S_
    qkv = torch.randn(B * N, 3 * D, generator=g) * 0.5
    if X3:
        from sam_pt_amd.pack import x3_rows
        qkv = x3_rows(qkv).to(dev)
    else:
        qkv = qkv.half().to(dev)
    rh = (torch.randn(2 * S_ - 1, hd, generator=g) * 0.1).to(dev)
    rw = (torch.randn(2 * S_ - 1, hd, generator=g) * 0.1).to(dev)
    out = torch.empty(B * N, 2 * D if X3 else D, dtype=torch.float16, device=dev)

    def call():
        if X3:
            return lib.sampt_vit_attention_x3(_lib.ptr(qkv), _lib.ptr(rh), _lib.ptr(rw), _lib.ptr(out), B, S_, heads, hd,
                                              _lib.stream_ptr())
        return lib.sampt_vit_attention_f16(_lib.ptr(qkv), _lib.ptr(rh), _lib.ptr(rw), _lib.ptr(out), B, S_, heads, hd, None, 0,
                                           _lib.stream_ptr())

    for _ in range(15):                      # steady state: the first ~10 launches run 5 - 10 % slower (clock ramp)
        _lib.check(call(), "attn")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        call()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 40 * 1e-3
    fl = 4.0 * B * heads * N * N * hd
    print(f"{'x3 ' if X3 else ''}{name:7s} B={B:4d} N={N:5d}  {t * 1e6:9.1f} us  {fl / t / 1e12:7.1f} TFLOP/s  {(4 * B * N * D * 2) / t / 1e12:6.2f} TB/s")
