#!/usr/bin/env python
"""The tracker encoder's 3 x 3 stride-1 convolutions over pre-split planes (8 frames of 576 x 1024): the halo-tiled kernel
(csrc/conv_halo_x3.hip) against the implicit-GEMM LDS-DMA kernel it replaces (sampt_conv_set_halo), HIP events, 10 + 20 launches,
and the largest difference between their outputs.   python tools/conv_halo_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd import _lib  # noqa: E402
from sam_pt_amd.pack import split_f16x3  # noqa: E402

SHAPES = [(8, 288, 512, 64, 64, 4), (8, 144, 256, 96, 96, 3), (8, 72, 128, 128, 128, 3), (8, 36, 64, 128, 128, 3),
          (8, 144, 256, 416, 256, 1)]          # n, H, W, Cin, Cout, launches per encoder pass of 8 frames
lib, dev = _lib.load(), torch.device("cuda:0")
tot = {0: 0.0, 1: 0.0, 2: 0.0, 8: 0.0}
for (n, H, W, ci, co, cnt) in SHAPES:
    g = torch.Generator().manual_seed(ci + co)
    x = torch.relu(torch.randn(n, H, W, ci, generator=g)).to(dev)
    xh = x.half()
    xhl = torch.stack([xh, (x - xh.float()).half()]).contiguous()
    w = torch.randn(co, 9 * ci, generator=g) * (2.0 / (co * 9)) ** 0.5
    whl, b = split_f16x3(w).to(dev), torch.randn(co, generator=g).to(dev)
    ys, ts = {}, {}
    for on in (0, 2, 8, 1):               # 0 implicit GEMM, 2 halo with 4-wave workgroups everywhere, 1 halo (8 waves from 96 channels)
        lib.sampt_conv_set_halo(on)
        y = torch.zeros(n, H, W, co, device=dev)
        call = lambda: lib.sampt_conv2d_nhwc(4, _lib.ptr(xhl), _lib.ptr(whl), _lib.ptr(b), _lib.ptr(y), n, H, W, ci, co, 3, 3, 1, 1,
                                             _lib.stream_ptr())
        for _ in range(10):
            _lib.check(call(), "conv")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts[on] = e0.elapsed_time(e1) / 20 * 1e3
        ys[on] = y
        tot[on] += ts[on] * cnt
    flop = 2.0 * n * H * W * co * 9 * ci
    d = max((ys[0] - ys[1]).abs().max().item(), (ys[0] - ys[2]).abs().max().item())
    print(f"{ci:4d}->{co:4d} {H:3d}x{W:3d}: implicit GEMM {ts[0]:8.1f} us ({flop / ts[0] / 1e6:6.1f} TFLOP/s)   halo/4w {ts[2]:8.1f} us   halo/plain tile order {ts[8]:8.1f} us   halo {ts[1]:8.1f} us "
          f"({flop / ts[1] / 1e6:6.1f} TFLOP/s fp32-equivalent)   max |difference| {d:.2e}  (|y| max {ys[0].abs().max().item():.2f})")
lib.sampt_conv_set_halo(1)
print(f"3 x 3 stride-1 convolutions per 8-frame encoder pass: implicit GEMM {tot[0] / 1e3:.2f} ms, halo/4w {tot[2] / 1e3:.2f} ms, halo/plain tile order {tot[8] / 1e3:.2f} ms, "
      f"halo {tot[1] / 1e3:.2f} ms")
