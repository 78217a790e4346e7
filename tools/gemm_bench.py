#!/usr/bin/env python
"""Correctness + micro-benchmark of the fp16 ViT GEMM through the C ABI (HIP events on the launching stream).

  python tools/gemm_bench.py [frames] [zeros] [nocheck] [x3]

``x3``: time the split-fp16 ("f16x3") variants of the same launches instead (x3 rows in, x3 rows / f32 out; correctness of
those is tests/test_gpu_kernels.py::test_gemm_x3); TFLOP/s are fp32-equivalent (2 M N K / t: each product costs 3 MFMAs).

Checks the kernel the dispatcher picks (256 x 256 8-phase persistent kernel for the big shapes) against a torch fp64
product on the GPU for every epilogue the encoder uses (bias / GELU / residual / f16 out / row scatter / row gather / odd M),
then times the four ViT-H block GEMMs in their real epilogue modes on random operands (zero-filled operands run faster:
DVFS, /opt/skills/guides/cdna_hip_programming.md rule 25)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
ZERO = "zeros" in sys.argv
X3 = "x3" in sys.argv
CHECK = "nocheck" not in sys.argv and not X3
P, S = _lib.ptr, _lib.stream_ptr
g = torch.Generator().manual_seed(0)


def run(dt, A, W, bias, res, C, act, rowmap=None, a_rowmap=None, M=None):
    M = A.shape[0] if M is None else M
    N, K = W.shape
    _lib.check(lib.sampt_gemm_ex(dt, P(A), P(W), P(bias), P(res), P(C), M, N, K, act, 1.0, P(rowmap), P(a_rowmap), 0, 0, S()),
               "gemm_ex")


def check(M, N, K, dt, act, use_res, scatter=False, gather=False, inplace=False):
    A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    rows_out = M
    rowmap = a_rowmap = None
    Asrc = A
    if scatter:   # destination rows: a permutation into a larger matrix, every 7th row dropped
        rows_out = M + 64
        perm = torch.randperm(rows_out, generator=g)[:M].int()
        perm[::7] = -1
        rowmap = perm.to(dev)
    if gather:
        a_rowmap = torch.randint(0, M, (M,), generator=g).int().to(dev)
        Asrc = A[a_rowmap.long()]
    res = torch.randn(rows_out, N, generator=g).to(dev) if use_res else None
    C = torch.full((rows_out, N), 7.0, device=dev, dtype=torch.float16 if dt == 2 else torch.float32)
    if inplace:                      # the encoder's residual stream: C is the residual (rows no tile writes keep their value)
        C, res = res.clone(), res.clone()
        run(dt, A, W, bias, C, C, act, rowmap, a_rowmap)
    else:
        run(dt, A, W, bias, res, C, act, rowmap, a_rowmap)
    torch.cuda.synchronize()
    ref = Asrc.double() @ W.double().t() + bias.double()
    if act == 2:
        ref = F.gelu(ref)
    full = res.double().clone() if inplace else torch.full((rows_out, N), 7.0, device=dev, dtype=torch.float64)
    if scatter:
        keep = rowmap >= 0
        full[rowmap[keep].long()] = ref[keep] + (res[rowmap[keep].long()].double() if use_res else 0)
    else:
        full = ref + (res.double() if use_res else 0)
    err = (C.double() - full).abs().max().item() / full.abs().max().item()
    tol = 1.5e-3 if dt == 2 else 2e-5
    tag = f"M={M} N={N} K={K} dt={dt} act={act} res={int(use_res)} scatter={int(scatter)} gather={int(gather)} inplace={int(inplace)}"
    print(f"check {tag}: max err / max |ref| = {err:.2e} {'OK' if err < tol else 'FAIL'}", flush=True)
    return err < tol


ok = True
if CHECK:
    for args in [(512, 256, 128, 1, 0, False), (2688, 1280, 1280, 1, 0, True), (2688, 3840, 1280, 2, 0, False),
                 (2688, 5120, 1280, 2, 2, False), (2688, 1280, 5120, 1, 0, True), (1000, 512, 256, 1, 2, True),
                 (4900, 3840, 1280, 2, 0, False, True, False), (4900, 1280, 1280, 1, 0, True, False, True),
                 (8 * 2688 + 77, 1280, 1280, 1, 0, True, True, True), (300, 256, 128, 2, 0, False),
                 (257, 768, 3072, 1, 0, True), (16384, 2304, 768, 2, 0, False),
                 (2688, 1280, 1280, 1, 0, True, False, False, True), (2688 + 77, 1280, 5120, 1, 0, True, True, True, True)]:
        ok &= check(*args)
    # run-to-run determinism and independence of the launch's other rows: the first 2688 rows of a big launch == a small one
    A = (torch.randn(8 * 2688, 1280, generator=g) * 0.5).half().to(dev)
    W = (torch.randn(3840, 1280, generator=g) / 36.0).half().to(dev)
    bias = torch.randn(3840, generator=g).to(dev)
    C1 = torch.empty(8 * 2688, 3840, device=dev, dtype=torch.float16)
    C2 = torch.empty(2688, 3840, device=dev, dtype=torch.float16)
    C3 = torch.empty_like(C1)
    run(2, A, W, bias, None, C1, 0)
    run(2, A[:2688].contiguous(), W, bias, None, C2, 0)
    run(2, A, W, bias, None, C3, 0)
    torch.cuda.synchronize()
    same = bool((C1[:2688] == C2).all()) and bool((C1 == C3).all())
    print("bitwise: batch rows == single-frame rows, run == rerun:", "OK" if same else "FAIL", flush=True)
    ok &= same
    print("ALL CHECKS", "OK" if ok else "FAILED", flush=True)

D = 1280
Ml, Mg = B * 2688, B * 4096
shapes = [("qkv  live", Ml, 3 * D, D, 2, 0, False), ("proj live", Ml, D, D, 1, 0, True), ("fc1  live", Ml, 4 * D, D, 2, 2, False),
          ("fc2  live", Ml, D, 4 * D, 1, 0, True), ("qkv  glob", Mg, 3 * D, D, 2, 0, False), ("proj glob", Mg, D, D, 1, 0, True),
          ("fc1  glob", Mg, 4 * D, D, 2, 2, False), ("fc2  glob", Mg, D, 4 * D, 1, 0, True), ("square 4096", 4096, 4096, 4096, 2, 0, False),
          ("square 8192", 8192, 8192, 8192, 2, 0, False)]
tot_f = tot_t = 0.0
for (name, M, N, K, dt, act, use_res) in shapes:
    A = (torch.randn(M, K, generator=g) * 0.5)
    W = (torch.randn(N, K, generator=g) / K ** 0.5)
    if X3:
        from sam_pt_amd.pack import F16X3_WSHIFT, x3_rows
        A, W = x3_rows(A).to(dev), x3_rows(W, F16X3_WSHIFT).to(dev)
    else:
        A, W = A.half().to(dev), W.half().to(dev)
    if ZERO:
        A.zero_(), W.zero_()
    bias = torch.zeros(N, device=dev)
    Cc = torch.zeros(M, 2 * N if (X3 and dt == 2) else N, device=dev, dtype=torch.float16 if dt == 2 else torch.float32)
    res = Cc if use_res else None            # in place, as the encoder's residual stream
    if X3:
        def run(dt, A, W, bias, res, C, act, M=M, N=N, K=K):   # noqa: F811
            _lib.check(lib.sampt_gemm_ex(4 if dt == 2 else 3, P(A), P(W), P(bias), P(res), P(C), M, N, K, act, 2.0 ** -8, None, None,
                                         0, 0, S()), "gemm_ex x3")
    for _ in range(15):      # steady state: the first ~10 launches of a shape run 5 - 10 % slower (clock ramp, cold Infinity Cache)
        run(dt, A, W, bias, res, Cc, act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    for _ in range(reps):
        run(dt, A, W, bias, res, Cc, act)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    fl = 2.0 * M * N * K
    if "square" not in name:
        tot_f += fl * (7 if "live" in name else 25) / 32.0   # ViT-H, 16:9 frames: 7 blocks on the live rows, 25 at full size
        tot_t += t * (7 if "live" in name else 25) / 32.0
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d} {'x3 ' if X3 else ''}out={'f16' if dt == 2 else 'f32'} act={act} res={int(use_res)} "
          f"{t * 1e6:9.1f} us  {fl / t / 1e12:7.1f} TFLOP/s", flush=True)
    del A, W, Cc
print(f"ViT-H block mix (7 live-row + 25 full-grid blocks): {tot_f / tot_t / 1e12:7.1f} TFLOP/s", flush=True)
sys.exit(0 if ok else 1)
