#!/usr/bin/env python
"""Run-to-run determinism of the halo convolution's fused InstanceNorm statistics at the tracker encoder's shapes: N launches each,
mean / rstd and the map compared bit for bit with the first.   python tools/probes/conv_stats_determinism.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from sam_pt_amd import _lib  # noqa: E402
from sam_pt_amd.pack import split_f16x3  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib, dev = _lib.load(), torch.device("cuda:0")
lib.sampt_conv_set_halo(MODE)
print("sampt_conv_set_halo", MODE)
P, S = _lib.ptr, _lib.stream_ptr
for (n, H, W, ci, co) in [(4, 192, 256, 64, 64), (3, 192, 256, 64, 64)]:
    g = torch.Generator().manual_seed(ci + co)
    x = torch.relu(torch.randn(n, H, W, ci, generator=g)).to(dev)
    xh = x.half()
    xhl = torch.stack([xh, (x - xh.float()).half()]).contiguous()
    whl = split_f16x3(torch.randn(co, 9 * ci, generator=g) * (2.0 / (co * 9)) ** 0.5).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    chunks = ((H + 15) // 16) * ((W + 15) // 16)
    ws = torch.empty(n * chunks * co * 2, dtype=torch.float64, device=dev)
    bad_mr = bad_y = bad_ws = 0
    ref = None
    for it in range(N):
        y = torch.empty(n, H, W, co, device=dev)
        mr = torch.empty(n, co, 2, device=dev)
        ws.fill_(float("nan"))
        _lib.check(lib.sampt_conv3x3_planes_instnorm_stats(P(xhl), P(whl), P(b), P(y), n, H, W, ci, co, 1e-5, P(mr), P(ws), ws.numel() * 8,
                                                           S()), "conv + stats")
        cur = (y.clone(), mr.clone(), ws.clone())
        if ref is None:
            ref = cur
            continue
        if not torch.equal(cur[0], ref[0]):
            bad_y += 1
            if bad_y <= 3:
                d = (cur[0] != ref[0]).nonzero()
                print("   map differs at", int(d.shape[0]), "elements; img", sorted(set(d[:, 0].tolist())), "y", sorted(set(d[:, 1].tolist())),
                      "x", sorted(set(d[:, 2].tolist())), "c", sorted(set(d[:, 3].tolist())),
                      " max |diff|", float((cur[0] - ref[0]).abs().max()), " |y| max", float(ref[0].abs().max()))
                i0, y0, x0 = int(d[0, 0]), int(d[:, 1].min()), int(d[:, 2].min())
                dd = (cur[0] - ref[0])[i0, y0:y0 + 4, x0:x0 + 16, 48:64]
                print("   diff[row 0, x 0..3, c 48..51]:", dd[0, :4, :4].tolist())
                print("   diff[rows 0..3, x 0, c 48]:", dd[:, 0, 0].tolist(), " per-channel std over the 64 pixels:", dd.reshape(64, 16).std(0)[:4].tolist())
        bad_mr += int(not torch.equal(cur[1], ref[1]))
        if not torch.equal(cur[2], ref[2]):
            bad_ws += 1
            if bad_ws == 1:
                d = (cur[2] != ref[2]).view(n, chunks, co, 2)
                idx = d.nonzero()
                print("   first differing partials (img, tile, channel, which):", idx[:6].tolist(), "of", int(d.sum()), " nan:",
                      int(torch.isnan(cur[2]).sum()))
    print(f"{ci:4d}->{co:4d} {H}x{W} n={n}: {N} launches, differing maps {bad_y}, statistics {bad_mr}, partial buffers {bad_ws}")
    bad_plain, ref = 0, None                                   # the same convolution without the statistics epilogue
    for it in range(N):
        y = torch.empty(n, H, W, co, device=dev)
        _lib.check(lib.sampt_conv2d_nhwc(4, P(xhl), P(whl), P(b), P(y), n, H, W, ci, co, 3, 3, 1, 1, S()), "conv")
        if ref is None:
            ref = y.clone()
        else:
            bad_plain += int(not torch.equal(y, ref))
    print(f"       without statistics: differing maps {bad_plain}")
