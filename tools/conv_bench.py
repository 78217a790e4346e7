"""Tracker-encoder convolution shapes (8 frames of 576x1024): exact fp32 MFMA (dtype 0) vs split-fp16 "f16x3" (dtype 3).
Usage (GPU box): python tools/conv_bench.py [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from sam_pt_amd import _lib  # noqa: E402
from sam_pt_amd.pack import split_f16x3  # noqa: E402

SHAPES = [  # n, H, W, Cin, Cout, k, stride, pad, count per encoder pass
    (8, 288, 512, 64, 64, 3, 1, 1, 4), (8, 288, 512, 64, 96, 3, 2, 1, 1), (8, 144, 256, 96, 96, 3, 1, 1, 3),
    (8, 144, 256, 96, 128, 3, 2, 1, 1), (8, 72, 128, 128, 128, 3, 1, 1, 3), (8, 72, 128, 128, 128, 3, 2, 1, 1),
    (8, 36, 64, 128, 128, 3, 1, 1, 3), (8, 144, 256, 416, 256, 3, 1, 1, 1), (8, 144, 256, 256, 128, 1, 1, 0, 1),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    lib, dev = _lib.load(), torch.device("cuda:0")
    tot = {0: 0.0, 3: 0.0}
    for (n, H, W, ci, co, k, st, pd, cnt) in SHAPES:
        g = torch.Generator().manual_seed(ci + co)
        x = torch.relu(torch.randn(n, H, W, ci, generator=g)).to(dev)
        w = (torch.randn(co, k * k * ci, generator=g) * (2.0 / (co * k * k)) ** 0.5)
        b = torch.randn(co, generator=g).to(dev)
        wd, whl = w.to(dev), split_f16x3(w).to(dev)
        OH, OW = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
        y = {0: torch.empty(n, OH, OW, co, device=dev), 3: torch.empty(n, OH, OW, co, device=dev)}
        flop = 2.0 * n * OH * OW * co * k * k * ci
        line = f"{ci:4d}->{co:4d} k{k} s{st} {OH:3d}x{OW:3d}  {flop / 1e9:7.1f} GF "
        for dt, wt in ((0, wd), (3, whl)):
            for _ in range(2):
                _lib.check(lib.sampt_conv2d_nhwc(dt, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(b), _lib.ptr(y[dt]), n, H, W, ci, co,
                                                 k, k, st, pd, _lib.stream_ptr()), "conv")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.sampt_conv2d_nhwc(dt, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(b), _lib.ptr(y[dt]), n, H, W, ci, co, k, k, st,
                                      pd, _lib.stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tot[dt] += ms * cnt
            line += f"| dtype {dt}: {ms:7.3f} ms {flop / ms / 1e9:6.1f} TF "
        err = ((y[3] - y[0]).abs().max() / y[0].abs().max()).item()
        print(line + f"| max diff {err:.1e}", flush=True)
    print(f"encoder pass (8 frames, convs with Cin%32==0): fp32 {tot[0]:.2f} ms, f16x3 {tot[3]:.2f} ms")


if __name__ == "__main__":
    main()
