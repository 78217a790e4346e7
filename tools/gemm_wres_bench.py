#!/usr/bin/env python
"""The mask decoder's image-side projections (24 frames x 4096 tokens): weights-resident kernel (csrc/gemm_x3_wres.hip) against the
tiled kernel it replaces (sampt_gemm_set_wres), HIP events, 10 + 20 launches.   python tools/gemm_wres_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sam_pt_amd import _lib  # noqa: E402
from sam_pt_amd.pack import split_f16x3  # noqa: E402

F_ = 24
SHAPES = [("kvq + pe", F_ * 4096, 384, 256, "mod", 0, 0, 2), ("i2t out + residual", F_ * 4096, 256, 128, "full", 0, 0, 2),
          ("final k / v", F_ * 4096, 128, 256, None, 0, 0, 2), ("upscale 0", F_ * 4096, 256, 256, None, 0, 64, 1),
          ("upscale 1 + GELU", 4 * F_ * 4096, 128, 64, None, 2, 128, 1)]      # name, M, N, K, res, act, shuf_g, launches per decoder pass
lib, dev = _lib.load(), torch.device("cuda:0")
tot = {0: 0.0, 1: 0.0}
for (name, M, N, K, res, act, sg, cnt) in SHAPES:
    g = torch.Generator().manual_seed(N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    whl = split_f16x3(torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    cout = N // 4 if sg else N
    b = torch.randn(cout, generator=g).to(dev)
    r = None if res is None else torch.randn(4096 if res == "mod" else M, N, generator=g).to(dev)
    rows = 4 * M if sg else M
    ys, ts = {}, {}
    for on in (0, 1):
        lib.sampt_gemm_set_wres(on)
        y = torch.zeros(rows, cout, device=dev)
        call = lambda: lib.sampt_gemm_x3_rows(_lib.ptr(A), _lib.ptr(whl), _lib.ptr(b), _lib.ptr(r) if r is not None else None,
                                              4096 if res == "mod" else 0, _lib.ptr(y), M, N, K, act, sg, _lib.stream_ptr())
        for _ in range(10):
            _lib.check(call(), "gemm_x3_rows")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts[on] = e0.elapsed_time(e1) / 20 * 1e3
        ys[on] = y
        tot[on] += ts[on] * cnt
    mb = (M * K * 4 + rows * cout * 4 + (0 if r is None else (M * N * 4 if res == "full" else 0))) / 1e6
    print(f"{name:20s} M {M:6d} N {N:3d} K {K:3d}: tiled {ts[0]:7.1f} us   weights-resident {ts[1]:7.1f} us  ({mb / ts[1]:5.2f} TB/s of A + C"
          f"{' + residual' if res == 'full' else ''}: {mb:.0f} MB)   bitwise equal: {torch.equal(ys[0], ys[1])}")
lib.sampt_gemm_set_wres(1)
print(f"image-side GEMMs per decoder pass of {F_} frames: tiled {tot[0] / 1e3:.2f} ms, weights-resident {tot[1] / 1e3:.2f} ms")
