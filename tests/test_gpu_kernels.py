"""GPU parity tests, kernel level: every HIP kernel family against a plain PyTorch fp32/fp64 reference of the same op
(or the oracle) on the same seeded inputs.  All calls go through the C ABI (ctypes)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.util import max_abs, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from sam_pt_amd import _lib
    return _lib.load()


def P(t):
    from sam_pt_amd import _lib
    return _lib.ptr(t)


def S():
    from sam_pt_amd import _lib
    return _lib.stream_ptr()


def ok(rc, what=""):
    from sam_pt_amd import _lib
    _lib.check(rc, what)


@pytest.mark.parametrize("M,N,K,act", [(300, 200, 64, 0), (64, 1040, 512, 0), (1, 32, 256, 1), (4096, 128, 256, 0),
                                         (130, 70, 520, 2), (16, 2048, 512, 2), (777, 513, 36, 0),
                                         # the thin kernel's shapes (K split over the waves of a workgroup): PIPS mixer
                                         # fc1 / fc2 / input projection at 8 chains, 48 chains, decoder token rows
                                         (64, 2048, 512, 2), (64, 512, 2048, 0), (64, 512, 520, 0), (384, 512, 2048, 2),
                                         (360, 256, 2048, 1), (17, 36, 68, 0), (129, 1040, 72, 2)])
def test_gemm_f32(lib, dev, M, N, K, act):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().t() * 0.5 + b.double()
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    ref = ref + R.double()
    Ad, Wd, bd, Rd = A.to(dev), W.to(dev), b.to(dev), R.to(dev)
    Cd = torch.empty(M, N, device=dev)
    ok(lib.sampt_gemm(0, P(Ad), P(Wd), P(bd), P(Rd), P(Cd), M, N, K, act, 0.5, S()), "gemm f32")
    assert rel_err(Cd, ref) < 2e-6


@pytest.mark.parametrize("M,N,K,dtype", [(256, 256, 128, 1), (4900, 3840, 1280, 2), (100, 72, 64, 1), (4096, 768, 768, 2),
                                           (300, 200, 192, 1), (129, 129, 64, 2), (1000, 1280, 320, 1),
                                           (4096, 5120, 1280, 2), (777, 320, 128, 1)])
def test_gemm_f16(lib, dev, M, N, K, dtype):
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    b = torch.randn(N, generator=g)
    ref = F.gelu(A.double() @ W.double().t() + b.double())
    Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)
    Cd = torch.empty(M, N, device=dev, dtype=torch.float16 if dtype == 2 else torch.float32)
    ok(lib.sampt_gemm(dtype, P(Ad), P(Wd), P(bd), None, P(Cd), M, N, K, 2, 1.0, S()), "gemm f16")
    assert rel_err(Cd.float(), ref) < (2e-3 if dtype == 2 else 2e-5 * K ** 0.5)


def _row_maps(M, g, drop=7):
    """A destination permutation with a few dropped rows (-1) and a gather permutation, as the ViT's window maps are."""
    dest = torch.randperm(M + 5, generator=g)[:M].to(torch.int32)
    dest[torch.randperm(M, generator=g)[:drop]] = -1
    return dest, torch.randperm(M, generator=g).to(torch.int32)


@pytest.mark.parametrize("M,N,K,dtype,act", [(512, 512, 256, 1, 0), (777, 1280, 1280, 1, 0), (1000, 768, 384, 2, 2),
                                               (4100, 1280, 5120, 1, 0), (300, 200, 192, 1, 0)])
def test_gemm_f16_epilogues(lib, dev, M, N, K, dtype, act):
    """sampt_gemm_ex on the 256 x 256 8-phase kernel's shapes (and one fallback shape) WITH everything the encoder's epilogues
    use at once: bias, in-place-style f32 residual at the row-mapped destination, row scatter (rowmap, dropped rows), row
    gather (a_rowmap), f32 / f16 output — against fp64."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = torch.randn(M, K, generator=g).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    b = torch.randn(N, generator=g)
    dest, src = _row_maps(M, g)
    R = torch.randn(M + 5, N, generator=g)
    acc = A[src.long()].double() @ W.double().t() + b.double()
    acc = F.gelu(acc) if act == 2 else acc
    use_res = dtype == 1
    ref = torch.full((M + 5, N), 7.0, dtype=torch.float64)
    keep = dest >= 0
    ref[dest[keep].long()] = acc[keep] + (R[dest[keep].long()].double() if use_res else 0.0)
    Ad, Wd, bd, Rd, dd, sd_ = A.to(dev), W.to(dev), b.to(dev), R.to(dev), dest.to(dev), src.to(dev)
    Cd = torch.full((M + 5, N), 7.0, device=dev, dtype=torch.float16 if dtype == 2 else torch.float32)
    ok(lib.sampt_gemm_ex(dtype, P(Ad), P(Wd), P(bd), P(Rd) if use_res else None, P(Cd), M, N, K, act, 1.0, P(dd), P(sd_), 0, 0,
                         S()), "gemm_ex f16")
    torch.cuda.synchronize()
    assert rel_err(Cd.float(), ref) < (2e-3 if dtype == 2 else 2e-5 * K ** 0.5)
    # residual broadcast (res_mod: the positional embedding of the patch GEMM) without row maps
    if use_res:
        mod = 100
        ref2 = A.double() @ W.double().t() + b.double() + R[:mod].double().repeat((M + mod - 1) // mod, 1)[:M]
        C2 = torch.empty(M, N, device=dev)
        ok(lib.sampt_gemm_ex(1, P(Ad), P(Wd), P(bd), P(Rd), P(C2), M, N, K, 0, 1.0, None, None, mod, 0, S()), "gemm_ex res_mod")
        assert rel_err(C2, ref2) < 2e-5 * K ** 0.5


@pytest.mark.parametrize("M,N,K,out_x3,act", [(512, 512, 256, 0, 0), (777, 1280, 1280, 0, 0), (1000, 768, 384, 1, 2),
                                                (4100, 1280, 5120, 0, 0), (300, 256, 192, 1, 0), (100, 72, 64, 0, 2),
                                                (129, 129, 64, 0, 0), (2050, 3840, 1280, 1, 0), (333, 160, 128, 0, 0)])
def test_gemm_x3(lib, dev, M, N, K, out_x3, act):
    """Split-fp16 GEMM (GemmP::x3: hi.hi + hi.lo + lo.hi on x3 rows; the "f16x3" precision of the image encoder): as close to
    the fp64 product as the exact f32 MFMA path, on the 8-phase kernel's shapes and on both fallback kernels, with the
    encoder's epilogues (bias, GELU, residual, row scatter / gather), f32 and x3-row outputs, and operands that only a
    saturating split keeps finite."""
    from sam_pt_amd.pack import F16X3_WSHIFT, x3_rows, x3_unrows
    g = torch.Generator().manual_seed(M + 7 * N + K)
    A = torch.randn(M, K, generator=g) * 1.7
    A[3, 5], A[M - 1, K - 1], A[0, 0] = 1.0e5, -9.0e4, 3e-6         # beyond the fp16 range: hi saturates, lo carries the rest
    W = torch.randn(N, K, generator=g) / K ** 0.5
    W[0, 0], W[1, 1] = 3e-6, -37.0
    b = torch.randn(N, generator=g)
    dest, src = _row_maps(M, g)
    R = torch.randn(M + 5, N, generator=g)
    acc = A[src.long()].double() @ W.double().t() + b.double()
    acc = F.gelu(acc) if act == 2 else acc
    use_res = not out_x3
    keep = dest >= 0
    ref = torch.full((M + 5, N), 7.0, dtype=torch.float64)
    ref[dest[keep].long()] = acc[keep] + (R[dest[keep].long()].double() if use_res else 0.0)
    Ax, Wx = x3_rows(A).to(dev), x3_rows(W, F16X3_WSHIFT).to(dev)
    assert rel_err(x3_unrows(Ax.cpu()), A) < 1e-6
    bd, Rd, dd, sd_ = b.to(dev), R.to(dev), dest.to(dev), src.to(dev)
    alpha = 1.0 / (1 << F16X3_WSHIFT)
    if out_x3:
        Cx = x3_rows(torch.full((M + 5, N), 7.0)).to(dev)
        ok(lib.sampt_gemm_ex(4, P(Ax), P(Wx), P(bd), None, P(Cx), M, N, K, act, alpha, P(dd), P(sd_), 0, 0, S()), "gemm x3 -> x3")
        got = x3_unrows(Cx.cpu())
    else:
        Cd = torch.full((M + 5, N), 7.0, device=dev)
        ok(lib.sampt_gemm_ex(3, P(Ax), P(Wx), P(bd), P(Rd), P(Cd), M, N, K, act, alpha, P(dd), P(sd_), 0, 0, S()), "gemm x3 -> f32")
        got = Cd
    # yardstick: the exact f32 MFMA GEMM on the same (gathered) operands
    A32, W32 = A[src.long()].contiguous().to(dev), W.to(dev)
    C32 = torch.empty(M, N, device=dev)
    ok(lib.sampt_gemm(0, P(A32), P(W32), P(bd), None, P(C32), M, N, K, act, 1.0, S()), "gemm f32")
    # Rows fed by the out-of-range operands are judged on their own scale: a saturated element is 65504 + lo with a lo that is
    # no longer small, so the dropped lo.lo term leaves it fp16-grade (2^-11 of that one product) — finite and sane, which is
    # what saturation is for.  Every other row is held to the exact f32 MFMA path's error.
    big = torch.zeros(M, dtype=torch.bool)
    big[(src == 3) | (src == M - 1)] = True
    for sel, floor, slack in ((big & keep, 3e-4, 2.0), (~big & keep, 2e-6, 2.0)):
        e32 = rel_err(C32.cpu()[sel], acc[sel])
        e3 = rel_err(got.cpu()[dest[sel].long()], ref[dest[sel].long()])
        assert e3 < max(floor, slack * e32), (e3, e32)
    assert torch.equal(got.cpu()[[i for i in range(M + 5) if i not in set(dest[keep].tolist())]].float(),
                       torch.full((M + 5 - int(keep.sum()), N), 7.0)), "rows outside the row map were written"
    # the device-side splitter writes the very rows the host packer does
    Ad = A.to(dev)
    Ay = torch.empty(M, 2 * K, dtype=torch.float16, device=dev)
    ok(lib.sampt_split_rows_x3(P(Ad), P(Ay), M, K, S()), "split_rows_x3")
    assert torch.equal(Ay.cpu(), Ax.cpu())


@pytest.mark.parametrize("n,H,W,Cin,Cout,k,s,p", [(2, 20, 28, 64, 96, 3, 2, 1), (1, 32, 48, 4, 64, 7, 2, 3),
                                                  (2, 17, 23, 96, 128, 1, 2, 0), (1, 16, 24, 416, 256, 3, 1, 1)])
def test_conv_f32(lib, dev, n, H, W, Cin, Cout, k, s, p):
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    bd = b.to(dev)
    y = torch.empty(ref.shape, device=dev)
    ok(lib.sampt_conv2d_nhwc(0, P(xd), P(wd), P(bd), P(y), n, H, W, Cin, Cout, k, k, s, p, S()), "conv f32")
    assert rel_err(y, ref) < 2e-6


@pytest.mark.parametrize("n,H,W,Cin,Cout,k,s,p", [(2, 20, 28, 64, 64, 3, 1, 1), (2, 20, 28, 64, 96, 3, 2, 1),
                                                  (2, 17, 23, 96, 128, 1, 2, 0), (1, 16, 24, 416, 256, 3, 1, 1),
                                                  (3, 9, 11, 128, 128, 3, 1, 1), (1, 40, 56, 96, 96, 3, 1, 1),
                                                  # halo-tiled kernel: whole tiles, one pixel, a ragged channel tile, three chunks
                                                  (2, 32, 48, 64, 64, 3, 1, 1), (1, 1, 1, 32, 64, 3, 1, 1), (1, 33, 17, 96, 100, 3, 1, 1),
                                                  (1, 18, 35, 128, 256, 3, 1, 1)])
def test_conv_f16x3(lib, dev, n, H, W, Cin, Cout, k, s, p):
    """Split-fp16 convolution (three fp16 MFMAs per fp32 product, csrc/conv_f16x3.hip): as close to the fp64 convolution
    as the exact-fp32 MFMA path is."""
    from sam_pt_amd.pack import split_f16x3
    g = torch.Generator().manual_seed(Cin * 3 + Cout)
    x = torch.relu(torch.randn(n, Cin, H, W, generator=g)) * 1.7
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cout * k * k)) ** 0.5
    w[0, 0, 0, 0], w[1 % Cout, 1, 0, 0] = 3e-6, -37.0                              # tiny and large weights
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    w2 = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    whl, wd, bd = split_f16x3(w2).to(dev), w2.to(dev), b.to(dev)
    y, y32 = torch.empty(ref.shape, device=dev), torch.empty(ref.shape, device=dev)
    ok(lib.sampt_conv2d_nhwc(3, P(xd), P(whl), P(bd), P(y), n, H, W, Cin, Cout, k, k, s, p, S()), "conv f16x3")
    ok(lib.sampt_conv2d_nhwc(0, P(xd), P(wd), P(bd), P(y32), n, H, W, Cin, Cout, k, k, s, p, S()), "conv f32")
    e3, e32 = rel_err(y, ref), rel_err(y32, ref)
    assert e3 < max(2e-6, 1.5 * e32), (e3, e32)
    # pre-split activations (dtype 4: the LDS-DMA kernel the tracker encoder runs): same products, same bar
    xh = xd.half()
    xhl = torch.stack([xh, (xd - xh.float()).half()]).contiguous()
    y4 = torch.full(ref.shape, 7.0, device=dev)
    ok(lib.sampt_conv2d_nhwc(4, P(xhl), P(whl), P(bd), P(y4), n, H, W, Cin, Cout, k, k, s, p, S()), "conv f16x3 planes")
    e4 = rel_err(y4, ref)
    assert e4 < max(2e-6, 1.5 * e32), (e4, e32)
    # ... and, where the halo-tiled kernel takes the launch (3 x 3, stride 1: csrc/conv_halo_x3.hip), the implicit-GEMM LDS-DMA kernel
    # it replaced must agree with it to the products' fp32 round-off
    if k == 3 and s == 1:
        try:
            ok(lib.sampt_conv_set_halo(0), "set_halo")
            y5 = torch.full(ref.shape, 7.0, device=dev)
            ok(lib.sampt_conv2d_nhwc(4, P(xhl), P(whl), P(bd), P(y5), n, H, W, Cin, Cout, k, k, s, p, S()), "conv f16x3 planes, implicit GEMM")
        finally:
            ok(lib.sampt_conv_set_halo(1), "set_halo")
        assert rel_err(y5, ref) < max(2e-6, 1.5 * e32)
        assert max_abs(y4, y5) < 5e-5 * max(1.0, float(ref.abs().max()))
    # unsupported shapes are refused, never silently computed another way
    assert lib.sampt_conv2d_nhwc(3, P(xd), P(whl), P(bd), P(y), n, H, W, Cin - 4, Cout, k, k, s, p, S()) == -3


@pytest.mark.parametrize("M,N,K,res,act,shuf_g", [
    (4 * 4096, 384, 256, "mod", 0, 0),        # fused K | V | Q' projection + projected positional embedding (engine_dec fused_proj)
    (4 * 4096 + 37, 256, 128, "full", 0, 0),  # image -> token block's output projection + residual stream; ragged last group
    (5 * 4096, 128, 256, None, 0, 0),         # final attention's k / v projections
    (4 * 4096, 256, 256, None, 0, 64),        # output_upscaling stage 0 as one GEMM over the four sub-pixels
    (16 * 4096, 128, 64, None, 2, 128),       # ... stage 1 with GELU
    (17000, 200, 64, "full", 0, 0),           # ragged column slice, K = 64 with the residual ring (two groups per loop body)
    (16500, 100, 128, None, 2, 0),
])
def test_gemm_x3_rows_weights_resident(lib, dev, M, N, K, res, act, shuf_g):
    """The decoder's image-side projections on the weights-resident kernel (csrc/gemm_x3_wres.hip): fp32-grade against the fp64
    product, and BITWISE the tiled kernel it replaces (same per-accumulator sequence of MFMAs, same epilogue order)."""
    from sam_pt_amd.pack import split_f16x3
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * 1.3
    A[5, 3], A[M - 1, K - 1] = 6.0e4, -3.0e-6                       # near the top of the fp16 range, and tiny
    w = torch.randn(N, K, generator=g) * (1.0 / K) ** 0.5
    cout = N // 4 if shuf_g else N
    b = torch.randn(cout, generator=g)
    P_ = 4096
    r = None if res is None else (torch.randn(P_ if res == "mod" else M, N, generator=g))
    ref = A.double() @ w.double().T
    if shuf_g:
        G_ = shuf_g
        F_ = M // (G_ * G_)
        ref = ref.view(F_, G_, G_, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(F_ * 4 * G_ * G_, cout)   # (f, y, dy, x, dx, c)
    ref = ref + b.double()
    if act == 2:
        ref = F.gelu(ref)
    if r is not None:
        ref = ref + (r.double().repeat(M // P_, 1) if res == "mod" else r.double())
    Ad, whl, bd = A.to(dev), split_f16x3(w).to(dev), b.to(dev)
    rd = None if r is None else r.to(dev)
    outs = []
    try:
        for on in (1, 0):
            ok(lib.sampt_gemm_set_wres(on), "set_wres")
            y = torch.full(ref.shape, 7.0, device=dev)
            ok(lib.sampt_gemm_x3_rows(P(Ad), P(whl), P(bd), P(rd) if rd is not None else None, P_ if res == "mod" else 0, P(y), M, N, K,
                                      act, shuf_g, S()), "gemm_x3_rows")
            outs.append(y)
    finally:
        ok(lib.sampt_gemm_set_wres(1), "set_wres")
    assert rel_err(outs[0], ref) < 2e-6
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(2, 32, 48, 64, 64), (1, 33, 17, 96, 100), (2, 18, 35, 128, 256), (3, 16, 16, 32, 128),
                                            (4, 192, 256, 64, 64), (4, 96, 128, 96, 96)])
def test_conv3x3_fused_instnorm_statistics(lib, dev, n, H, W, Cin, Cout):
    """The halo convolution's epilogue sums the following InstanceNorm's statistics (GemmP::in_part): mean and 1 / sqrt(var + eps)
    per (image, channel) against fp64 statistics of the very map the kernel wrote — ragged tiles, two column tiles, a non-zero mean."""
    from sam_pt_amd.pack import split_f16x3
    g = torch.Generator().manual_seed(n * H + Cout)
    x = torch.relu(torch.randn(n, H, W, Cin, generator=g)) * 1.7
    w = torch.randn(Cout, 9 * Cin, generator=g) * (2.0 / (Cout * 9)) ** 0.5
    b = torch.randn(Cout, generator=g) * 3.0                                   # |mean| >> std in some channels: E[x^2] - mean^2 cancels
    xd = x.to(dev)
    xh = xd.half()
    xhl = torch.stack([xh, (xd - xh.float()).half()]).contiguous()
    whl, bd = split_f16x3(w).to(dev), b.to(dev)
    y = torch.full((n, H, W, Cout), 7.0, device=dev)
    mr = torch.full((n, Cout, 2), 7.0, device=dev)
    chunks = ((H + 15) // 16) * ((W + 15) // 16)
    ws = torch.empty(n * chunks * Cout * 2, dtype=torch.float64, device=dev)
    ok(lib.sampt_conv3x3_planes_instnorm_stats(P(xhl), P(whl), P(bd), P(y), n, H, W, Cin, Cout, 1e-5, P(mr), P(ws), ws.numel() * 8, S()),
       "conv + stats")
    y2 = torch.empty_like(y)
    ok(lib.sampt_conv2d_nhwc(4, P(xhl), P(whl), P(bd), P(y2), n, H, W, Cin, Cout, 3, 3, 1, 1, S()), "conv")
    assert torch.equal(y, y2)                                                  # the statistics do not disturb the convolution
    mr2 = torch.full((n, Cout, 2), 7.0, device=dev)                            # and again: bit for bit (fixed summation order)
    ok(lib.sampt_conv3x3_planes_instnorm_stats(P(xhl), P(whl), P(bd), P(y2), n, H, W, Cin, Cout, 1e-5, P(mr2), P(ws), ws.numel() * 8, S()),
       "conv + stats")
    assert torch.equal(y, y2) and torch.equal(mr, mr2)
    yd = y.double().reshape(n, H * W, Cout)
    mean, var = yd.mean(1), yd.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    assert (mr[:, :, 0].double() - mean).abs().max().item() < 2e-6 * max(1.0, mean.abs().max().item())
    assert ((mr[:, :, 1].double() - rstd) / rstd).abs().max().item() < 5e-6


@pytest.mark.parametrize("n,H,W", [(2, 64, 96), (1, 45, 71), (1, 32, 32), (3, 36, 130), (4, 384, 512)])   # (last: 768 tiles on 512 workgroups)
def test_conv_stem7x7_split_fp16(lib, dev, n, H, W):
    """The tracker encoder's stem as split-fp16 products (csrc/conv_stem_x3.hip): as close to the fp64 convolution as the exact-fp32
    MFMA path, borders and ragged tiles included; fused InstanceNorm statistics against fp64 statistics of the written map."""
    g = torch.Generator().manual_seed(H * W)
    img = torch.randint(0, 256, (n, 3, H, W), generator=g).float()
    x = 2.0 * (img / 255.0) - 1.0
    w = torch.randn(64, 3, 7, 7, generator=g) * (2.0 / (64 * 49)) ** 0.5
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=3).permute(0, 2, 3, 1)
    OH, OW = ref.shape[1:3]
    x4 = torch.cat([x, torch.zeros(n, 1, H, W)], 1).permute(0, 2, 3, 1).contiguous().to(dev)
    w4 = torch.cat([w, torch.zeros(64, 1, 7, 7)], 1).permute(0, 2, 3, 1).contiguous().to(dev)      # [64][7][7][4]
    bd = b.to(dev)
    y = torch.full(ref.shape, 7.0, device=dev)
    mr = torch.full((n, 64, 2), 7.0, device=dev)
    chunks = ((OH + 15) // 16) * ((OW + 15) // 16)
    ws = torch.empty(n * chunks * 64 * 2, dtype=torch.float64, device=dev)
    ok(lib.sampt_conv_stem7x7(P(x4), P(w4), P(bd), P(y), n, H, W, 1e-5, P(mr), P(ws), ws.numel() * 8, S()), "stem")
    y32 = torch.empty(ref.shape, device=dev)
    ok(lib.sampt_conv2d_nhwc(0, P(x4), P(w4.reshape(64, -1)), P(bd), P(y32), n, H, W, 4, 64, 7, 7, 2, 3, S()), "conv f32")
    e3, e32 = rel_err(y, ref), rel_err(y32, ref)
    assert e3 < max(2e-6, 1.5 * e32), (e3, e32)
    yd = y.double().reshape(n, OH * OW, 64)
    mean, rstd = yd.mean(1), 1.0 / torch.sqrt(yd.var(1, unbiased=False) + 1e-5)
    assert (mr[:, :, 0].double() - mean).abs().max().item() < 2e-6 * max(1.0, mean.abs().max().item())
    assert ((mr[:, :, 1].double() - rstd) / rstd).abs().max().item() < 5e-6
    y2 = torch.full(ref.shape, 7.0, device=dev)                               # without statistics: the same map
    ok(lib.sampt_conv_stem7x7(P(x4), P(w4), P(bd), P(y2), n, H, W, 1e-5, None, None, 0, S()), "stem, no statistics")
    assert torch.equal(y, y2)
    mr2 = torch.full((n, 64, 2), 7.0, device=dev)                             # and again: bit for bit (fixed summation order)
    ok(lib.sampt_conv_stem7x7(P(x4), P(w4), P(bd), P(y2), n, H, W, 1e-5, P(mr2), P(ws), ws.numel() * 8, S()), "stem")
    assert torch.equal(y, y2) and torch.equal(mr, mr2)


@pytest.mark.parametrize("F_", [4, 7])
def test_gemm_x3_rows_fused_upscaling_tails(lib, dev, F_):
    """output_upscaling with its LayerNorm2d + GELU and the mask's dot product done in the weights-resident GEMMs' epilogues
    (csrc/gemm_x3_wres.hip epi = 1 / 2): bit for bit the separate kernels' results, and fp32-grade against fp64."""
    from sam_pt_amd.pack import split_f16x3
    g = torch.Generator().manual_seed(F_)
    G_, P_ = 64, 4096
    keys = torch.randn(F_ * P_, 256, generator=g)
    w0, b0 = torch.randn(4 * 64, 256, generator=g) / 16, torch.randn(64, generator=g)
    lnw, lnb = 1.0 + 0.2 * torch.randn(64, generator=g), 0.3 * torch.randn(64, generator=g)
    w1, b1 = torch.randn(4 * 32, 64, generator=g) / 8, torch.randn(32, generator=g)
    hyp = torch.randn(F_, 32, generator=g)
    kd, w0d, w1d = keys.to(dev), split_f16x3(w0).to(dev), split_f16x3(w1).to(dev)
    b0d, b1d, lnwd, lnbd, hypd = b0.to(dev), b1.to(dev), lnw.to(dev), lnb.to(dev), hyp.to(dev)
    # separate kernels
    mid = torch.empty(4 * F_ * P_, 64, device=dev)
    ok(lib.sampt_gemm_x3_rows(P(kd), P(w0d), P(b0d), None, 0, P(mid), F_ * P_, 256, 256, 0, G_, S()), "stage 0")
    ok(lib.sampt_layernorm(P(mid), P(lnwd), P(lnbd), P(mid), 4 * F_ * P_, 64, 1e-6, 0, 2, S()), "LayerNorm2d + GELU")
    up1 = torch.empty(16 * F_ * P_, 32, device=dev)
    ok(lib.sampt_gemm_x3_rows(P(mid), P(w1d), P(b1d), None, 0, P(up1), 4 * F_ * P_, 128, 64, 2, 2 * G_, S()), "stage 1")
    low = torch.empty(F_, 16 * P_, device=dev)
    ok(lib.sampt_sam_mask_dot(P(up1), P(hypd), 32, P(low), F_, 16 * P_, 32, S()), "mask dot")
    # fused tails
    mid2 = torch.full((4 * F_ * P_, 64), 7.0, device=dev)
    ok(lib.sampt_gemm_x3_rows_epi(P(kd), P(w0d), P(b0d), None, 0, P(mid2), F_ * P_, 256, 256, 2, G_, 1, P(lnwd), P(lnbd), 1e-6, 0, S()),
       "stage 0 + LayerNorm2d + GELU")
    assert torch.equal(mid, mid2)
    low2 = torch.full((F_, 16 * P_), 7.0, device=dev)
    ok(lib.sampt_gemm_x3_rows_epi(P(mid2), P(w1d), P(b1d), None, 0, P(low2), 4 * F_ * P_, 128, 64, 2, 2 * G_, 2, P(hypd), None, 0.0, 32, S()),
       "stage 1 + mask dot")
    assert torch.equal(low, low2)
    # fp64 reference of the whole tail for one frame
    f = F_ - 1
    x = keys[f * P_:(f + 1) * P_].double() @ w0.double().T                                   # (P, 4 * 64): columns (dy, dx, c)
    x = x.view(G_, G_, 2, 2, 64).permute(0, 2, 1, 3, 4).reshape(4 * P_, 64) + b0.double()
    x = F.gelu(F.layer_norm(x, (64,), lnw.double(), lnb.double(), 1e-6))
    y = (x @ w1.double().T).view(2 * G_, 2 * G_, 2, 2, 32).permute(0, 2, 1, 3, 4).reshape(16 * P_, 32) + b1.double()
    ref = F.gelu(y) @ hyp[f].double()
    assert rel_err(low2[f], ref) < 5e-6
    # shapes without a fused tail are refused, not computed another way
    assert lib.sampt_gemm_x3_rows_epi(P(kd), P(w0d), P(b0d), None, 0, P(mid2), 2 * P_, 256, 256, 2, G_, 1, P(lnwd), P(lnbd), 1e-6, 0, S()) == -3


def test_gemm_x3_rows_fused_layernorm_tail(lib, dev):
    """keys = LayerNorm(keys + attn_out W^T + b) of the decoder's image -> token block in one kernel (gemm_x3_wres.hip epi = 3):
    bit for bit the projection followed by sampt_layernorm, in place, with a ragged last group of rows."""
    from sam_pt_amd.pack import split_f16x3
    g = torch.Generator().manual_seed(11)
    M = 5 * 4096 + 5
    att, keys = torch.randn(M, 128, generator=g), torch.randn(M, 256, generator=g) * 2.0
    w, b = torch.randn(256, 128, generator=g) / 11, torch.randn(256, generator=g)
    lnw, lnb = 1.0 + 0.2 * torch.randn(256, generator=g), 0.3 * torch.randn(256, generator=g)
    ad, whl, bd, lnwd, lnbd = att.to(dev), split_f16x3(w).to(dev), b.to(dev), lnw.to(dev), lnb.to(dev)
    k1 = keys.to(dev).clone()
    ok(lib.sampt_gemm_x3_rows(P(ad), P(whl), P(bd), P(k1), 0, P(k1), M, 256, 128, 0, 0, S()), "projection + residual (in place)")
    ok(lib.sampt_layernorm(P(k1), P(lnwd), P(lnbd), P(k1), M, 256, 1e-5, 0, 0, S()), "LayerNorm")
    k2 = keys.to(dev).clone()
    ok(lib.sampt_gemm_x3_rows_epi(P(ad), P(whl), P(bd), P(k2), 0, P(k2), M, 256, 128, 0, 0, 3, P(lnwd), P(lnbd), 1e-5, 0, S()), "fused")
    assert torch.equal(k1, k2)
    ref = F.layer_norm(keys.double() + att.double() @ w.double().T + b.double(), (256,), lnw.double(), lnb.double(), 1e-5)
    assert rel_err(k2, ref) < 3e-6


def test_conv_f16(lib, dev):
    n, H, W, Cin, Cout = 2, 16, 16, 256, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half()
    ref = F.conv2d(x.double(), w.double(), None, padding=1).permute(0, 2, 3, 1)
    y = torch.empty(ref.shape, device=dev)
    xd, wd = x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev)  # keep alive
    ok(lib.sampt_conv2d_nhwc(1, P(xd), P(wd), None, P(y), n, H, W, Cin, Cout, 3, 3, 1, 1, S()), "conv f16")
    assert rel_err(y, ref) < 1e-4


@pytest.mark.parametrize("C_", [64, 96, 128, 256])
def test_instance_norm(lib, dev, C_):
    n, H, W = 3, 37, 52
    g = torch.Generator().manual_seed(C_)
    x = torch.randn(n, C_, H, W, generator=g) * 3 + 1
    skip = torch.randn(n, C_, H, W, generator=g)
    ref = F.relu(F.relu(F.instance_norm(x.double(), eps=1e-5)) + skip.double()).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    sd = skip.permute(0, 2, 3, 1).contiguous().to(dev)
    nb = lib.sampt_instance_norm_workspace_bytes(n, H * W, C_)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    ok(lib.sampt_instance_norm_nhwc(P(xd), n, H * W, C_, 1e-5, 1, P(sd), P(ws), nb, S()), "instnorm")
    assert max_abs(xd, ref) < 5e-6


@pytest.mark.parametrize("D,f16,act", [(64, 0, 2), (256, 0, 0), (512, 0, 0), (768, 1, 0), (1280, 1, 0), (4, 0, 2), (16, 0, 0),
                                       (1280, 2, 0), (768, 2, 0), (64, 2, 0), (96, 2, 2), (256, 3, 0), (768, 3, 2)])
def test_layernorm(lib, dev, D, f16, act):
    M = 333
    g = torch.Generator().manual_seed(D)
    x = torch.randn(M, D, generator=g) * 2 + 0.5
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ref = F.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-6)
    ref = F.gelu(ref) if act == 2 else ref
    y = torch.empty((2, M, D) if f16 == 3 else (M, 2 * D if f16 == 2 else D), device=dev, dtype=torch.float16 if f16 else torch.float32)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)  # keep the device copies alive across the async launch
    ok(lib.sampt_layernorm(P(xd), P(wd), P(bd), P(y), M, D, 1e-6, f16, act, S()), "layernorm")
    if f16 == 3:                                  # two fp16 planes (the halo convolution's operand): hi = fp16(result), hi + lo the fp32 result
        y32 = torch.empty(M, D, device=dev)
        ok(lib.sampt_layernorm(P(xd), P(wd), P(bd), P(y32), M, D, 1e-6, 0, act, S()), "layernorm f32")
        assert torch.equal(y[0], y32.half()) and torch.equal(y[1], (y32 - y[0].float()).half())
        assert max_abs(y[0].float() + y[1].float(), ref) < 1e-5
        assert lib.sampt_layernorm(P(xd), P(wd), P(bd), P(y), M, 96, 1e-6, 3, act, S()) == -3     # planes: the vectorised kernel's widths only
    elif f16 == 2:                                # x3 rows: hi + lo is the fp32 result
        from sam_pt_amd.pack import x3_unrows
        assert max_abs(x3_unrows(y.cpu()), ref) < 1e-5
    else:
        assert max_abs(y.float(), ref) < (2e-2 if f16 else 1e-5)


@pytest.mark.parametrize("align", [0, 1])
@pytest.mark.parametrize("sh,sw,dh,dw", [(36, 64, 144, 256), (40, 52, 20, 26), (16, 24, 16, 24), (9, 13, 21, 30)])
def test_resize_bilinear(lib, dev, align, sh, sw, dh, dw):
    n, C_ = 2, 8
    x = torch.randn(n, C_, sh, sw, generator=torch.Generator().manual_seed(sh))
    ref = F.interpolate(x, (dh, dw), mode="bilinear", align_corners=bool(align)).permute(0, 2, 3, 1)
    y = torch.zeros(n, dh, dw, 12, device=dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    ok(lib.sampt_resize_bilinear_nhwc(P(xd), n, sh, sw, C_, P(y), dh, dw, 12, 4,
                                      align, S()), "resize")
    assert max_abs(y[..., 4:], ref) < 2e-6
    assert float(y[..., :4].abs().max()) == 0.0


def test_avgpool(lib, dev):
    x = torch.randn(2, 128, 16, 24, generator=torch.Generator().manual_seed(1))
    ref = F.avg_pool2d(x, 2, stride=2).permute(0, 2, 3, 1)
    y = torch.empty(2, 8, 12, 128, device=dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    ok(lib.sampt_avgpool2x2_nhwc(P(xd), 2, 16, 24, 128, P(y), S()), "avgpool")
    assert max_abs(y, ref) < 1e-6


def test_corr_sample_vs_oracle(lib, dev):
    """Fused local correlation + 7x7 sampler == CorrBlock.corr + CorrBlock.sample (pips.py:364-407), including
    points near / outside the border (zeros padding) and integer-valued coordinates."""
    from oracle import pips_ref as O
    from sam_pt_amd import _lib
    g = torch.Generator().manual_seed(3)
    S_, n, H0, W0 = 8, 6, 32, 48
    fm = torch.randn(S_, 128, H0, W0, generator=g)
    pyr = O.build_pyramid(fm)
    ffeats = torch.randn(S_, n, 128, generator=g)
    coords = torch.rand(S_, n, 2, generator=g) * torch.tensor([W0 - 1.0, H0 - 1.0])
    coords[:, 0] = torch.tensor([0.3, 0.2])          # near the top-left corner
    coords[:, 1] = torch.tensor([W0 + 2.5, H0 - 0.5])  # partly outside
    coords[:, 2] = torch.tensor([10.0, 7.0])          # exactly integral
    ref = O.sample_corr(O.corr_volumes(pyr, ffeats), coords)           # S,N,196
    pyr_d = [p.permute(0, 2, 3, 1).contiguous().to(dev) for p in pyr]
    fidx = torch.arange(S_, dtype=torch.int32, device=dev).repeat(n, 1).contiguous()    # [n][S]
    ff_d = ffeats.permute(1, 0, 2).contiguous().to(dev)                # [n][S][128]
    out = torch.empty(n, S_, 196, device=dev)
    co_d = coords.contiguous().to(dev)
    ok(lib.sampt_corr_sample_f32(_lib.ptr_array(pyr_d), H0, W0, P(fidx), S_, n, P(ff_d), P(co_d), P(out), S()), "corr_sample")
    assert max_abs(out.permute(1, 0, 2), ref) < 2e-5 * float(ref.abs().max())


def _ref_vit_attention(qkv, rel_h, rel_w, B, S_, heads, hd, dtype=torch.float64):
    N, D = S_ * S_, heads * hd
    qkv, rel_h, rel_w = qkv.to(dtype), rel_h.to(dtype), rel_w.to(dtype)
    q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, B * heads, N, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    idx = torch.arange(S_)[:, None] - torch.arange(S_)[None, :] + (S_ - 1)
    Rh, Rw = rel_h[idx], rel_w[idx]
    rq = q.reshape(B * heads, S_, S_, hd)
    attn = attn.view(-1, S_, S_, S_, S_) + torch.einsum("bhwc,hkc->bhwk", rq, Rh)[..., None] \
        + torch.einsum("bhwc,wkc->bhwk", rq, Rw)[:, :, :, None, :]
    attn = attn.view(-1, N, N).softmax(-1)
    return (attn @ v).view(B, heads, N, hd).permute(0, 2, 1, 3).reshape(B * N, D)


# (19 windows / 3 x 3 (frame, head) pairs: two / one full groups of 8 units of the XCD-aware work order plus a tail, common.h
#  flash_wg_decode)
@pytest.mark.parametrize("B,S_,heads,hd", [(2, 64, 2, 80), (3, 14, 4, 80), (3, 64, 3, 64), (5, 14, 2, 64), (2, 16, 2, 32),
                                            (4, 6, 2, 32), (19, 14, 2, 80), (11, 16, 2, 32)])
def test_vit_flash_attention(lib, dev, B, S_, heads, hd):
    g = torch.Generator().manual_seed(S_ * hd)
    N, D = S_ * S_, heads * hd
    qkv = (torch.randn(B * N, 3 * D, generator=g) * 1.5).half()
    rel_h = torch.randn(2 * S_ - 1, hd, generator=g) * 0.3
    rel_w = torch.randn(2 * S_ - 1, hd, generator=g) * 0.3
    ref = _ref_vit_attention(qkv, rel_h, rel_w, B, S_, heads, hd)
    out = torch.empty(B * N, D, device=dev, dtype=torch.float16)
    qd, hd_, wd_ = qkv.to(dev), rel_h.to(dev), rel_w.to(dev)
    ok(lib.sampt_vit_attention_f16(P(qd), P(hd_), P(wd_), P(out), B, S_, heads, hd, None, 0, S()), "flash")
    assert max_abs(out.float(), ref) < 6e-3 * float(ref.abs().max())


@pytest.mark.parametrize("B,S_,heads,hd", [(2, 64, 2, 80), (3, 14, 4, 80), (1, 64, 2, 64), (5, 14, 2, 64), (2, 16, 2, 32),
                                            (4, 6, 2, 32), (19, 14, 2, 80), (11, 16, 2, 32)])
def test_vit_flash_attention_x3(lib, dev, B, S_, heads, hd):
    """The split-fp16 attention kernel (precision "f16x3"): fp32-grade against the fp64 attention of the same fp32 q / k / v —
    three orders of magnitude tighter than the fp16 kernel's bar — including scores large enough that softmax is peaked."""
    from sam_pt_amd.pack import x3_rows, x3_unrows
    g = torch.Generator().manual_seed(S_ * hd + 1)
    N, D = S_ * S_, heads * hd
    qkv = torch.randn(B * N, 3 * D, generator=g) * 1.5
    rel_h = torch.randn(2 * S_ - 1, hd, generator=g) * 0.3
    rel_w = torch.randn(2 * S_ - 1, hd, generator=g) * 0.3
    ref = _ref_vit_attention(qkv, rel_h, rel_w, B, S_, heads, hd)
    e32 = max_abs(_ref_vit_attention(qkv, rel_h, rel_w, B, S_, heads, hd, dtype=torch.float32), ref)   # the same in plain fp32
    qx = x3_rows(qkv).to(dev)
    out = torch.empty(B * N, 2 * D, device=dev, dtype=torch.float16)
    hd_, wd_ = rel_h.to(dev), rel_w.to(dev)
    ok(lib.sampt_vit_attention_x3(P(qx), P(hd_), P(wd_), P(out), B, S_, heads, hd, S()), "flash x3")
    # fp32-grade = within a small factor of what plain fp32 arithmetic leaves on the same inputs: with scores of magnitude ~30
    # (q, k ~ N(0, 1.5^2), peaked softmax) the fp32 rounding of the score itself dominates both (measured 3.1x at 64 x 64
    # tokens / head dim 80, 1 - 2x elsewhere; the fp16 kernel is at 300x)
    e3 = max_abs(x3_unrows(out.cpu()), ref)
    assert e3 < max(4e-6 * float(ref.abs().max()), 5.0 * e32), (e3, e32)


@pytest.mark.parametrize("in_h,in_w,oh,ow", [(576, 1024, 576, 1024), (576, 1024, 480, 854), (1024, 683, 300, 200)])
def test_postprocess_and_bbox(lib, dev, in_h, in_w, oh, ow):
    low = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(oh)) * 2 - 1.0
    ref = F.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)[..., :in_h, :in_w]
    ref = F.interpolate(ref, (oh, ow), mode="bilinear", align_corners=False)[0, 0]
    out = torch.empty(oh, ow, device=dev)
    low_d = low.to(dev)
    ok(lib.sampt_postprocess_masks(P(low_d), 256, 1024, in_h, in_w, P(out), oh, ow, S()), "postprocess")
    assert max_abs(out, ref) < 1e-5
    bb = torch.zeros(5, dtype=torch.int32, device=dev)
    nb = lib.sampt_bbox_workspace_bytes(oh, ow)
    bws = torch.empty(nb, dtype=torch.uint8, device=dev)
    ok(lib.sampt_bbox_from_logits(P(out), oh, ow, P(bb), P(bws), nb, S()), "bbox")
    m = out.cpu() > 0
    yx = m.nonzero()
    exp = [int(yx[:, 1].min()), int(yx[:, 0].min()), int(yx[:, 1].max()), int(yx[:, 0].max()), int(m.sum())]
    assert bb.cpu().tolist() == exp


def test_resize_logits_and_index_masks(lib, dev):
    """VOS post-processing kernels vs torch: F.interpolate(align_corners=False) incl. -inf maps, and
    argmax(softmax(cat(bg=0, logits)))."""
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(3, 4, 36, 64, generator=g) * 2
    logits[1, 2] = -float("inf")                                   # a rejected mask (sam_pt.py:834-835)
    ref = F.interpolate(logits, size=(30, 53), mode="bilinear", align_corners=False)
    ld = logits.to(dev)
    out = torch.empty(3, 4, 30, 53, device=dev)
    ok(lib.sampt_resize_logits(P(ld), 12, 36, 64, P(out), 30, 53, S()), "resize_logits")
    o, r = out.cpu(), ref
    fin = torch.isfinite(r)
    assert (torch.isfinite(o) == fin).all() and max_abs(o[fin], r[fin]) < 1e-5
    bg = torch.zeros(1, 4, 36, 64)
    exp = torch.softmax(torch.cat([bg, logits]), dim=0).argmax(dim=0).to(torch.uint8)
    idx = torch.empty(4, 36, 64, dtype=torch.uint8, device=dev)
    ok(lib.sampt_index_masks(P(ld), 3, 4 * 36 * 64, P(idx), S()), "index_masks")
    assert torch.equal(idx.cpu(), exp)
    # NaN logits (the resized -inf map has 0 * -inf = NaN where its bilinear taps are clamped at the border): the
    # reference's softmax row is all-NaN there and argmax gives 0 = background
    # (up-scaling clamps the source coordinate at the border, which makes one interpolation weight exactly 0)
    ref_up = F.interpolate(logits, size=(45, 80), mode="bilinear", align_corners=False)
    assert torch.isnan(ref_up).any()
    out_up = torch.empty(3, 4, 45, 80, device=dev)
    ok(lib.sampt_resize_logits(P(ld), 12, 36, 64, P(out_up), 45, 80, S()), "resize_logits")
    assert torch.equal(torch.isnan(out_up.cpu()), torch.isnan(ref_up))
    exp2 = torch.softmax(torch.cat([torch.zeros(1, 4, 45, 80), ref_up]), dim=0).argmax(dim=0).to(torch.uint8)
    idx2 = torch.empty(4, 45, 80, dtype=torch.uint8, device=dev)
    ok(lib.sampt_index_masks(P(out_up), 3, 4 * 45 * 80, P(idx2), S()), "index_masks")
    assert torch.equal(idx2.cpu(), exp2)


def test_vos_index_masks_overrides(lib, dev):
    """Evaluator overrides fused into the index-mask kernel (vos_eval/eval.py:318-326) vs the torch formula."""
    from sam_pt_amd.dist import index_masks
    g = torch.Generator().manual_seed(5)
    M, T, H, W = 3, 5, 24, 40
    logits = torch.randn(M, T, H, W, generator=g) * 3
    logits[1, 2] = -float("inf")
    qt = torch.tensor([0, 2, 4])
    gt = (torch.rand(M, H, W, generator=g) > 0.6).float()
    gt[2] = gt[0]                                                                  # overlapping GT: the first object wins
    ref = index_masks(logits, qt, gt)
    got = index_masks(logits.to(dev), qt, gt.to(dev))
    assert torch.equal(got.cpu(), ref)
    ref2 = index_masks(logits, qt, None)
    got2 = index_masks(logits.to(dev), qt, None)
    assert torch.equal(got2.cpu(), ref2) and not torch.equal(ref, ref2)
    lnan = logits.clone()
    lnan[0, 1, :3] = float("nan")          # before object 1's query frame the override hides nothing of object 0's NaNs
    lnan[1, 0, 5:9] = float("nan")         # ... but these are replaced by -1e8 (t < qt[1]) and must not matter
    ref3 = index_masks(lnan, qt, None)
    assert torch.equal(index_masks(lnan.to(dev), qt, None).cpu(), ref3) and (ref3[1, :3] == 0).all()
    assert (ref[:2] != 2).all() and (ref[:4] != 3).all()                           # nothing before the query frame


def _vos_resized_prob64(logits, qt, gt, out_hw):
    """softmax -> bilinear resize (align_corners=False) of the evaluator's tail (vos_eval/eval.py:326, 340-356) in float64.  The
    source indices and interpolation weights are computed in float32 exactly as F.interpolate and the kernel compute them
    (scale * (dst + 0.5) - 0.5, clamped at 0), then promoted: the only freedom left to an fp32 implementation is rounding."""
    M, T, H, W = logits.shape
    lg = logits.double().clone()
    if qt is not None:
        for m in range(M):
            t = int(qt[m])
            lg[m, :t] = -1e8
            if gt is not None:
                lg[m, t] = torch.where(gt[m] > 0, 1e8, -1e8).double()
    prob = torch.softmax(torch.cat([torch.zeros((1, T, H, W), dtype=torch.float64), lg], dim=0), dim=0)     # (M+1,T,H,W)

    def axis(n_in, n_out):
        scale = torch.tensor(n_in, dtype=torch.float32) / torch.tensor(n_out, dtype=torch.float32)
        src = (scale * (torch.arange(n_out, dtype=torch.float32) + 0.5) - 0.5).clamp(min=0)
        i0 = src.floor().long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        lam = (src - i0.float()).double()
        return i0, i1, lam

    y0, y1, ly = axis(H, out_hw[0])
    x0, x1, lx = axis(W, out_hw[1])
    rows = prob[:, :, y0] * (1 - ly)[:, None] + prob[:, :, y1] * ly[:, None]
    return rows[..., x0] * (1 - lx) + rows[..., x1] * lx                                                    # (M+1,T,oh,ow)


def test_vos_index_masks_resized(lib, dev):
    """softmax -> bilinear resize of the probabilities -> argmax (vos_eval/eval.py:326, 340-356) fused in one kernel, with
    and without the query-frame overrides, up- and down-scaling.  EXACT wherever the answer is determined: against a float64
    evaluation of the same formula the index must equal the argmax at every pixel whose two best probabilities differ by more
    than 1e-5 (fp32 rounding of exp / the four-tap sum cannot bridge that), and at the remaining pixels — provable ties: the
    +-1e8 overrides make probabilities exactly 0 / 1, so interpolation weights of 0.5 tie two labels — it must be one of the
    tied labels.  The torch fp32 formula on the CPU is held to the same statement (it differs from the kernel only there)."""
    from sam_pt_amd.dist import index_masks
    g = torch.Generator().manual_seed(9)
    M, T, H, W = 3, 4, 36, 64
    logits = torch.randn(M, T, H, W, generator=g) * 4
    logits[2, 1] = -1e30
    qt = torch.tensor([0, 1, 3])
    gt = (torch.rand(M, H, W, generator=g) > 0.5).float()
    TIE = 1e-5
    n_tie = n_all = 0
    for out_hw in ((30, 53), (72, 128), (36, 100)):
        for q, m in ((None, None), (qt, None), (qt, gt)):
            ref = index_masks(logits, q, m, out_hw=out_hw)
            got = index_masks(logits.to(dev), q, None if m is None else m.to(dev), out_hw=out_hw).cpu()
            assert got.shape == (T,) + out_hw
            p64 = _vos_resized_prob64(logits, q, m, out_hw)                       # (M+1,T,oh,ow)
            top2 = p64.topk(2, dim=0).values
            decided = (top2[0] - top2[1]) > TIE
            want = p64.argmax(dim=0)
            for name, idx in (("kernel", got), ("torch formula", ref)):
                idx = idx.long()
                assert torch.equal(idx[decided], want[decided]), (name, out_hw, int((idx[decided] != want[decided]).sum()))
                p_idx = p64.gather(0, idx[None])[0]
                assert bool((p_idx[~decided] >= top2[0][~decided] - TIE).all()), (name, out_hw, "a tie resolved to a non-tied label")
            n_tie += int((~decided).sum())
            n_all += decided.numel()
            # wherever the kernel and the torch formula disagree it is one of those ties
            assert bool((~decided)[got != ref].all()), out_hw
    print(f"\n[vos resized] {n_tie} of {n_all} output pixels are provable ties (two labels within {TIE} in float64)")
    assert n_tie < 0.02 * n_all


def _mha_ref(q, k, v, heads, nk=None):
    """q (F,Nq,D), k/v (F,Nk,D) -> softmax(q k^T / sqrt(hd)) v per head, fp64; nk[f] = valid keys of item f."""
    F_, Nq, D = q.shape
    hd = D // heads
    out = torch.empty(F_, Nq, D, dtype=torch.float64)
    for f in range(F_):
        n = k.shape[1] if nk is None else int(nk[f])
        qq = q[f].double().view(Nq, heads, hd).permute(1, 0, 2)
        kk = k[f, :n].double().view(n, heads, hd).permute(1, 0, 2)
        vv = v[f, :n].double().view(n, heads, hd).permute(1, 0, 2)
        a = torch.softmax(qq @ kk.transpose(1, 2) / hd ** 0.5, dim=-1)
        out[f] = (a @ vv).permute(1, 0, 2).reshape(Nq, D)
    return out


@pytest.mark.parametrize("kind,hd,Nq,Nk", [(1, 16, 300, 100), (1, 16, 4096, 128), (1, 16, 1000, 129), (1, 16, 517, 307),
                                           (1, 16, 260, 1707), (0, 32, 128, 128), (0, 32, 307, 307), (0, 32, 50, 1707),
                                           (0, 16, 307, 4096), (0, 16, 7, 200)])
def test_decoder_attention_kernels(lib, dev, kind, hd, Nq, Nk):
    """The mask decoder's fp32 attention kernels against an fp64 reference, incl. prompts beyond 128 tokens (chunked
    token-key attention) and ragged batches (per-item key counts)."""
    heads, F_ = 8, 2
    g = torch.Generator().manual_seed(Nq * 31 + Nk)
    D = heads * hd
    q, k, v = (torch.randn(F_, n, D, generator=g) for n in (Nq, Nk, Nk))
    out = torch.empty(F_, Nq, D, device=dev)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    ok(lib.sampt_attention_f32(kind, P(qd), P(kd), P(vd), P(out), F_, Nq, Nk, heads, hd, None, S()), "attention")
    assert max_abs(out, _mha_ref(q, k, v, heads)) < 2e-5
    nk = torch.tensor([Nk, max(1, Nk // 2 + 3)], dtype=torch.int32)
    ok(lib.sampt_attention_f32(kind, P(qd), P(kd), P(vd), P(out), F_, Nq, Nk, heads, hd, P(nk.to(dev)), S()), "attention")
    assert max_abs(out, _mha_ref(q, k, v, heads, nk)) < 2e-5


@pytest.mark.parametrize("F_,Nq,Nk", [(24, 15, 4096), (2, 16, 4096), (1, 7, 4096), (3, 21, 1000), (1, 300, 4096), (40, 9, 257),
                                      (2, 5, 333)])
def test_token_to_image_attention_kernel(lib, dev, F_, Nq, Nk):
    """The split-key token -> image attention (8 heads x 16) vs an fp64 reference: full and ragged key ranges, several
    query blocks, splits with and without the merge launch."""
    heads, D = 8, 128
    g = torch.Generator().manual_seed(F_ * 1000 + Nq * 10 + Nk)
    q, k, v = (torch.randn(F_, n, D, generator=g) for n in (Nq, Nk, Nk))
    q = q * 3.0                                                            # peaked softmax rows as well as flat ones
    out = torch.full((F_, Nq, D), float("nan"), device=dev)
    n = C.c_size_t()
    ok(lib.sampt_attention_t2i_workspace_bytes(F_, Nq, Nk, C.byref(n)), "t2i workspace")
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)                          # (kept alive: the call only takes pointers)
    ok(lib.sampt_attention_t2i_f32(P(qd), P(kd), P(vd), P(out), F_, Nq, Nk, P(ws), ws.numel(), S()), "t2i attention")
    assert max_abs(out, _mha_ref(q, k, v, heads)) < 2e-5


@pytest.mark.parametrize("nb,L,time_attn", [(36, 8, True), (8, 36, False), (8, 100, False), (8, 300, False), (5, 8, True)])
def test_cotracker_attention_kernel(lib, dev, nb, L, time_attn):
    """CoTracker's token-group attention read straight from packed qkv rows (8 heads x 48) vs an fp64 reference."""
    heads, hd, S_ = 8, 48, 8
    D = heads * hd
    g = torch.Generator().manual_seed(nb * 13 + L)
    npts = nb if time_attn else L
    qkv = torch.randn(npts * S_, 3 * D, generator=g)                   # rows = point * S + frame
    x = qkv.view(npts, S_, 3, D)
    grp = x if time_attn else x.transpose(0, 1)                        # (groups, tokens, 3, D)
    ref = _mha_ref(grp[:, :, 0], grp[:, :, 1], grp[:, :, 2], heads)    # (groups, tokens, D)
    ref = ref if time_attn else ref.transpose(0, 1)
    out = torch.empty(npts * S_, D, device=dev)
    bs, ts = (S_, 1) if time_attn else (1, S_)
    ok(lib.sampt_cotracker_attention_f32(P(qkv.to(dev)), P(out), nb, L, bs, ts, heads, hd, S()), "cot attention")
    assert max_abs(out.view(npts, S_, D), ref.reshape(npts, S_, D)) < 2e-5


def test_kmedoids_device_equals_host_restatement(lib, dev):
    """csrc/kmedoids.hip against sam_pt_amd.query_points.kmedoids_alternate (the host restatement of sklearn_extra's KMedoids that
    sam_pt/utils/query_points.py:62-99 calls), BIT-IDENTICAL on 50 seeded masks: the fp64 row sums (numpy's pairwise summation
    order), the medoids the alternating iterations converge to, and the query points ``extract_kmedoid_points`` returns."""
    import numpy as np
    from sam_pt_amd import query_points as Q
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:240, 0:320]
    for case in range(50):
        mask = np.zeros((240, 320), dtype=bool)
        for _ in range(int(rng.integers(1, 4))):                  # a few random ellipses: blobs, holes, thin and tiny masks
            cy, cx = rng.uniform(20, 220), rng.uniform(20, 300)
            ry, rx = rng.uniform(2, 70), rng.uniform(2, 90)
            mask |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        if case % 7 == 3:
            mask &= (yy + xx) % 3 != 0                            # perforated: many exactly equidistant pixels
        mt = torch.from_numpy(mask.astype(np.float32))
        K = int(rng.choice([1, 2, 4, 8, 16]))
        if int(mask.sum()) < K:
            continue
        px = mt.nonzero().float()
        px = px[torch.randperm(len(px), generator=torch.Generator().manual_seed(case))[:1800]]
        n = len(px)
        # row sums: np.sum(D, axis=1) bit for bit
        X = px.numpy().astype(np.float64)
        D = np.sqrt(np.maximum(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1), 0.0))
        xy = px.contiguous().to(dev)
        sums = torch.empty(n, dtype=torch.float64, device=dev)
        ok(lib.sampt_kmedoids_rowsums_f64(P(xy), n, P(sums), S()), "rowsums")
        assert np.array_equal(sums.cpu().numpy(), D.sum(axis=1)), f"case {case}: row sums differ"
        want = Q.kmedoids_alternate(px.numpy(), K)
        got = Q.kmedoids_alternate_device(px, K, dev)
        assert np.array_equal(got, want), f"case {case} (n={n}, K={K}): medoids {got} != {want}"
        # and through the public function, same RNG consumption on both paths
        torch.manual_seed(100 + case)
        a = Q.extract_kmedoid_points(mt, K)
        torch.manual_seed(100 + case)
        b = Q.extract_kmedoid_points(mt, K, device=dev)
        assert torch.equal(a, b)


def _seeded_mask_and_image(case, rng, H=240, W=320):
    """A few random ellipses (blobs, holes, thin and tiny masks) + a textured uint8 RGB image with blocks, gradients and noise
    (plenty of corners, flat areas and exact value ties)."""
    import numpy as np
    yy, xx = np.mgrid[0:H, 0:W]
    mask = np.zeros((H, W), dtype=bool)
    for _ in range(int(rng.integers(1, 4))):
        cy, cx = rng.uniform(20, H - 20), rng.uniform(20, W - 20)
        ry, rx = rng.uniform(2, 70), rng.uniform(2, 90)
        mask |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
    if case % 7 == 3:
        mask &= (yy + xx) % 3 != 0
    if case % 11 == 5:
        mask[:] = False
        mask[int(rng.integers(0, H - 3)):, int(rng.integers(0, W - 3)):][:3, :4] = True      # fewer than 10 pixels at all
    img = np.zeros((H, W, 3), np.float64)
    b = int(rng.integers(6, 40))
    img += (((yy // b) + (xx // b)) % 2)[..., None] * rng.uniform(40, 160)                  # checkerboard: exact ties
    img += (yy[..., None] * rng.uniform(0, 0.3, 3) + xx[..., None] * rng.uniform(0, 0.3, 3))
    if case % 3:
        img += rng.normal(0, rng.uniform(0.5, 12), (H, W, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    if case % 5 == 0:
        img[:, :, :] = img[:, :, :1]                                                       # gray image
    return mask, img


def test_erode_device_equals_host_restatement(lib, dev):
    """csrc/corners.hip erosion (separable) against sam_pt_amd.query_points._erode (cv2.erode with a k x k kernel of ones,
    query_points.py:165-194), bit-identical for k = 0 (OpenCV's default 3 x 3), 1 (identity), even, odd and large k, masks touching
    the border (pixels outside the image never erode)."""
    import numpy as np
    from sam_pt_amd import query_points as Q
    rng = np.random.default_rng(11)
    for case in range(12):
        mask, _ = _seeded_mask_and_image(case, rng)
        if case % 4 == 0:
            mask[:, :5] = True
            mask[-7:, :] = True
        for k in (0, 1, 2, 3, 4, 7, 16, 33):
            want = Q._erode(mask.astype(np.uint8), k)
            got = Q.erode_device(torch.from_numpy(mask.astype(np.float32)), k, dev).cpu().numpy()
            assert np.array_equal(got, want), (case, k)


def test_shi_tomasi_device_equals_host_restatement(lib, dev):
    """csrc/corners.hip against the numpy restatement of cv2.cvtColor / cv2.erode / cv2.goodFeaturesToTrack in
    sam_pt_amd/query_points.py (what sam_pt/utils/query_points.py:102-162 calls), BIT-IDENTICAL on 50 seeded (image, mask)
    pairs: which erosion of the 6 % / 2 % / 1 % / none cascade is kept, the eroded mask's pixel count and bounding box, and the
    selected corners in order — through the device entry point and through the public ``extract_corner_points`` (top-up with
    k-medoid points included, same RNG consumption on both paths)."""
    import numpy as np
    from sam_pt_amd import query_points as Q
    rng = np.random.default_rng(5)
    n_with_corners = n_topped_up = 0
    for case in range(50):
        mask, img = _seeded_mask_and_image(case, rng)
        if not mask.any():
            continue
        mt = torch.from_numpy(mask.astype(np.float32))
        it = torch.from_numpy(img).permute(2, 0, 1).contiguous()
        n = int(rng.choice([1, 2, 3, 5, 8, 16]))
        # the host cascade, step by step
        eroded, k_used = None, -1
        px = mt.nonzero().float()
        diam = torch.norm(px.max(0)[0] - px.min(0)[0]).item()
        for pct in (0.06, 0.02, 0.01):
            if eroded is None or eroded.sum() < 10:
                eroded, k_used = Q.erode_mask_proportional_to_its_furthest_points_distance(mt, pct), int(diam * pct)
        if eroded.sum() < 10:
            eroded, k_used = mt, -1
        k_used = 3 if k_used == 0 else k_used
        epx = eroded.nonzero().float()
        ediam = torch.norm(epx.max(0)[0] - epx.min(0)[0]).item()
        want = Q.good_features_to_track(Q._rgb_to_gray_u8(img), n, 0.001, ediam / n, eroded.numpy().astype(np.uint8))
        got, info = Q.shi_tomasi_device(it, mt, n, dev)
        assert info["k"] == k_used and info["eroded_pixels"] == int(eroded.sum()), (case, info, k_used, int(eroded.sum()))
        assert len(want) <= info["candidates"] <= eroded.numel() // 4 + 64   # the greedy rounds ran over the compacted list (corners.hip)
        assert info["eroded_bbox"] == [int(epx[:, 0].min()), int(epx[:, 0].max()), int(epx[:, 1].min()), int(epx[:, 1].max())]
        assert got.shape == want.shape and np.array_equal(got.numpy(), want), f"case {case} (n={n}): {got.tolist()} != {want.tolist()}"
        n_with_corners += len(want) > 0
        n_topped_up += len(want) < n
        torch.manual_seed(300 + case)
        a = Q.extract_corner_points(it, mt, n)
        torch.manual_seed(300 + case)
        b = Q.extract_corner_points(it.to(dev), mt, n, device=dev)
        assert torch.equal(a, b), case
    assert n_with_corners >= 30 and n_topped_up >= 3, (n_with_corners, n_topped_up)       # both regimes were exercised
    # the min-eigenvalue map itself, on the last image: same bits as the numpy evaluation
    eig = Q.corner_min_eigen_val(Q._rgb_to_gray_u8(img))
    assert np.isfinite(eig).all() and eig.max() > 0


@pytest.mark.parametrize("precision,S_,heads,hd,nwx,nwy,gh,gw", [(1, 14, 2, 80, 3, 2, 22, 36), (2, 14, 2, 80, 3, 2, 22, 36),
                                                                   (1, 6, 2, 32, 3, 3, 16, 16), (2, 6, 2, 32, 3, 3, 16, 16),
                                                                   (1, 14, 2, 64, 5, 3, 42, 64)])
def test_vit_window_attention_padding_from_bias_row(lib, dev, precision, S_, heads, hd, nwx, nwy, gh, gw):
    """Windowed blocks with SAM's zero padding (App. A-3: pad after norm1 -> a padded token's qkv is the bias): the kernels fetch
    padded keys / values from the bias row and never touch the padded rows of the qkv matrix (filled with NaN here) — same result
    as attention over a matrix whose padded rows hold the bias, on every real token."""
    from sam_pt_amd.pack import x3_rows, x3_unrows
    g = torch.Generator().manual_seed(S_ * hd + nwx)
    frames, N, D = 2, S_ * S_, heads * hd
    B = frames * nwx * nwy
    qkv = torch.randn(B * N, 3 * D, generator=g) * 1.2
    bias = torch.randn(3 * D, generator=g)
    rel_h = torch.randn(2 * S_ - 1, hd, generator=g) * 0.3
    rel_w = torch.randn(2 * S_ - 1, hd, generator=g) * 0.3
    w = torch.arange(B) % (nwx * nwy)
    iy, ix = torch.arange(N) // S_, torch.arange(N) % S_
    pad = (((w // nwx) * S_)[:, None] + iy[None] >= gh) | (((w % nwx) * S_)[:, None] + ix[None] >= gw)      # (B, N)
    assert bool(pad.any()) and not bool(pad.all())
    if precision == 1:
        qkv, bias = qkv.half().float(), bias.half().float()
    filled = torch.where(pad.reshape(-1, 1), bias[None], qkv)
    ref = _ref_vit_attention(filled, rel_h, rel_w, B, S_, heads, hd)
    poisoned = torch.where(pad.reshape(-1, 1), torch.full_like(qkv, float("nan")), qkv)
    if precision == 1:
        qd, bd = poisoned.half().to(dev), bias.half().to(dev)
        out = torch.empty(B * N, D, device=dev, dtype=torch.float16)
    else:
        qd = x3_rows(torch.nan_to_num(poisoned, nan=0.0))
        qd[pad.reshape(-1)] = float("nan")
        qd, bd = qd.to(dev), x3_rows(bias[None]).view(-1).to(dev)
        out = torch.empty(B * N, 2 * D, device=dev, dtype=torch.float16)
    hd_, wd_ = rel_h.to(dev), rel_w.to(dev)
    ok(lib.sampt_vit_window_attention(precision, P(qd), P(hd_), P(wd_), P(out), frames, S_, heads, hd, P(bd), nwx, nwy, gh, gw,
                                      S()), "window attention")
    got = out.float().cpu() if precision == 1 else x3_unrows(out.cpu())
    real = ~pad.reshape(-1)
    assert bool(torch.isfinite(got[real]).all())
    tol = 6e-3 if precision == 1 else 1e-5
    assert max_abs(got[real], ref[real]) < tol * float(ref.abs().max())


# ---- fused MLP-Mixer block of the PIPS window (csrc/pips_mixer.hip; pips.py:96-128) -------------------------------------
def _mixer_block_inputs(nseq, seed):
    g = torch.Generator().manual_seed(seed)
    R = nseq * 8
    x = torch.randn(R, 512, generator=g) * 1.5 + 0.3
    w = dict(lnw=1 + 0.2 * torch.randn(512, generator=g), lnb=0.1 * torch.randn(512, generator=g),
             w1=torch.randn(2048, 512, generator=g) / 512 ** 0.5, b1=0.1 * torch.randn(2048, generator=g),
             w2=torch.randn(512, 2048, generator=g) / 2048 ** 0.5, b2=0.1 * torch.randn(512, generator=g),
             tlnw=1 + 0.2 * torch.randn(512, generator=g), tlnb=0.1 * torch.randn(512, generator=g),
             tw1=torch.randn(32, 8, generator=g) / 8 ** 0.5, tb1=0.1 * torch.randn(32, generator=g),
             tw2=torch.randn(8, 32, generator=g) / 32 ** 0.5, tb2=0.1 * torch.randn(8, generator=g))
    return x, w


@pytest.mark.parametrize("nseq,slices", [(8, 8), (8, 16), (8, 32), (3, 8), (1, 32), (24, 8), (5, 16)])
def test_pips_mix_mlp_slabs(lib, dev, nseq, slices):
    """Channel MLP of a mixer block as hidden slices: the slabs must sum to fc2(gelu(fc1(LN(x)))) (fp64 reference) and every
    slab must be its own slice's product (a permuted slice order would still sum right)."""
    x, w = _mixer_block_inputs(nseq, 100 + nseq + slices)
    R = nseq * 8
    xd = x.double()
    y = F.layer_norm(xd, (512,), w["lnw"].double(), w["lnb"].double(), 1e-5)
    h = F.gelu(y @ w["w1"].double().t() + w["b1"].double())
    hs = 2048 // slices
    ref = torch.stack([h[:, s * hs:(s + 1) * hs] @ w["w2"].double()[:, s * hs:(s + 1) * hs].t() for s in range(slices)])
    d = {k: v.to(dev) for k, v in w.items()}
    part = torch.full((slices, R, 512), float("nan"), device=dev)
    x_d = x.to(dev)                                     # (named: a temporary's block could be handed out again before the launch)
    ok(lib.sampt_pips_mix_mlp_f32(P(x_d), P(d["lnw"]), P(d["lnb"]), P(d["w1"]), P(d["b1"]), P(d["w2"]), P(part), nseq,
                                  slices, S()), "mix_mlp")
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()
    scale = ref.abs().max().item()
    assert (part.cpu().double() - ref).abs().max().item() < 3e-6 * max(scale, 1.0)
    assert rel_err(part.sum(0), ref.sum(0)) < 2e-6


@pytest.mark.parametrize("nseq,slices,mode", [(8, 8, 0), (8, 0, 0), (8, 16, 1), (3, 32, 0), (1, 8, 1), (24, 8, 0), (5, 0, 1)])
def test_pips_mix_reduce(lib, dev, nseq, slices, mode):
    """Slab sum + bias + residual, then token mixing (mode 0) or final LayerNorm + token mean (mode 1), against fp64 torch."""
    x, w = _mixer_block_inputs(nseq, 200 + nseq + slices + mode)
    R = nseq * 8
    g = torch.Generator().manual_seed(7)
    part = torch.randn(max(slices, 1), R, 512, generator=g) * 0.3
    xp = x.double() + ((part[:slices].double().sum(0) + w["b2"].double()) if slices else 0.0)
    y = F.layer_norm(xp, (512,), w["tlnw"].double(), w["tlnb"].double(), 1e-5)
    if mode == 0:
        ys = y.view(nseq, 8, 512)                                             # token mixing: Conv1d(k = 1) over the 8 tokens
        hid = F.gelu(torch.einsum("ot,ntc->noc", w["tw1"].double(), ys) + w["tb1"].double()[None, :, None])
        mixed = torch.einsum("to,noc->ntc", w["tw2"].double(), hid) + w["tb2"].double()[None, :, None]
        ref = xp + mixed.reshape(R, 512)
        out = torch.full((R, 512), float("nan"), device=dev)
    else:
        ref = y.view(nseq, 8, 512).mean(1)
        out = torch.full((nseq, 512), float("nan"), device=dev)
    d = {k: v.to(dev) for k, v in w.items()}
    part_d, x_d = part.to(dev), x.to(dev)               # (named: two temporaries would share one recycled block)
    ok(lib.sampt_pips_mix_reduce_f32(P(part_d) if slices else None, slices, P(d["b2"]) if slices else None, P(x_d),
                                     nseq, mode, P(d["tlnw"]), P(d["tlnb"]), P(d["tw1"]), P(d["tb1"]), P(d["tw2"]), P(d["tb2"]),
                                     P(out), S()), "mix_reduce")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 2e-6


def test_pips_mixer_fused_equals_four_launch_blocks(lib, dev):
    """A clip's chained windows through both mixer paths (sampt_pips_set_mixer): same trajectories to fp32 round-off."""
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.weights import init_pips_state_dict
    from tests.util import disc_queries, synthetic_clip
    frames, centres = synthetic_clip(T=12, H=128, W=256, seed=5)
    q = disc_queries(centres, n_pos=5, r=9.0)[None]
    outs = []
    try:
        for fused in (0, 1):
            ok(lib.sampt_pips_set_mixer(fused, 32), "set_mixer")
            trk = PipsPointTracker(state_dict=init_pips_state_dict(72)).to(dev)
            tr, vi = trk(frames.to(dev)[None], q.to(dev))
            outs.append((tr.cpu(), vi.cpu()))
    finally:
        ok(lib.sampt_pips_set_mixer(2, 16), "set_mixer")       # the library's default
    assert max_abs(outs[0][0], outs[1][0]) < 2e-3, "trajectories of the two mixer paths differ"
    assert (outs[0][0].round() == outs[1][0].round()).all()
    assert (outs[0][1] == outs[1][1]).all()


# ---- split-fp16 generation of the mixer block (csrc/pips_mixer_x3.hip) ---------------------------------------------------------
def _xop_to_rows(xop, R):
    """Operand images [frag][ks][plane][lane][8] halves -> (hi + lo) rows [R][512] as float64 (the layout k_pips_mix_pre writes)."""
    nf = (R + 15) // 16
    v = xop.view(nf, 16, 2, 4, 16, 8).double()                 # [frag][ks][plane][lq][lr][e]
    rows = (v[:, :, 0] + v[:, :, 1]).permute(0, 3, 1, 2, 4)    # [frag][lr][ks][lq][e]
    return rows.reshape(nf * 16, 512)[:R]


@pytest.mark.parametrize("nseq,slices", [(8, 16), (8, 0), (3, 32), (1, 16), (24, 16), (5, 0)])
def test_pips_mix_pre(lib, dev, nseq, slices):
    """Slab sum + bias + residual + token mixing -> x'' (fp32) and the operand images of 2^6 LayerNorm2(x'') (fp16 hi + lo)."""
    x, w = _mixer_block_inputs(nseq, 300 + nseq + slices)
    R = nseq * 8
    g = torch.Generator().manual_seed(11)
    part = torch.randn(max(slices, 1), R, 512, generator=g) * 0.3
    xp = x.double() + ((part[:slices].double().sum(0) + w["b2"].double()) if slices else 0.0)
    y = F.layer_norm(xp, (512,), w["tlnw"].double(), w["tlnb"].double(), 1e-5)
    ys = y.view(nseq, 8, 512)
    hid = F.gelu(torch.einsum("ot,ntc->noc", w["tw1"].double(), ys) + w["tb1"].double()[None, :, None])
    mixed = torch.einsum("to,noc->ntc", w["tw2"].double(), hid) + w["tb2"].double()[None, :, None]
    xpp = xp + mixed.reshape(R, 512)
    y2 = F.layer_norm(xpp, (512,), w["lnw"].double(), w["lnb"].double(), 1e-5)
    d = {k: v.to(dev) for k, v in w.items()}
    part_d, x_d = part.to(dev), x.to(dev)
    xout = torch.full((R, 512), float("nan"), device=dev)
    xop = torch.zeros(lib.sampt_pips_mix_xop_halves(nseq), dtype=torch.float16, device=dev)
    ok(lib.sampt_pips_mix_pre_f32(P(part_d) if slices else None, slices, P(d["b2"]) if slices else None, P(x_d), nseq,
                                  P(d["tlnw"]), P(d["tlnb"]), P(d["tw1"]), P(d["tb1"]), P(d["tw2"]), P(d["tb2"]), P(d["lnw"]),
                                  P(d["lnb"]), P(xout), P(xop), S()), "mix_pre")
    torch.cuda.synchronize()
    assert rel_err(xout, xpp) < 2e-6
    got = _xop_to_rows(xop.cpu(), R) / 64.0
    assert (got - y2).abs().max().item() < 4e-6 * max(1.0, y2.abs().max().item())


@pytest.mark.parametrize("nseq,slices", [(8, 16), (8, 32), (3, 16), (1, 32), (24, 16), (9, 32)])
def test_pips_mix_mlp_x3_slabs(lib, dev, nseq, slices):
    """fc1 -> GELU -> fc2 over hidden slices as 3-term split-fp16 products from the packed weight stream: every slab against
    the fp64 product of its own slice, at fp32 grade."""
    from sam_pt_amd.pack import pips_mixer_x3_stream
    x, w = _mixer_block_inputs(nseq, 400 + nseq + slices)
    R = nseq * 8
    y = F.layer_norm(x.double(), (512,), w["lnw"].double(), w["lnb"].double(), 1e-5)
    h = F.gelu(y @ w["w1"].double().t() + w["b1"].double())
    hs = 2048 // slices
    ref = torch.stack([h[:, s * hs:(s + 1) * hs] @ w["w2"].double()[:, s * hs:(s + 1) * hs].t() for s in range(slices)])
    # operand images of the input, built on the host the way k_pips_mix_pre writes them
    nf = (R + 15) // 16
    ys = torch.zeros(nf * 16, 512)
    ys[:R] = (y * 64.0).float()
    hi = ys.half()
    lo = (ys - hi.float()).half()
    img = torch.stack([hi, lo]).view(2, nf, 16, 16, 4, 8).permute(1, 3, 0, 4, 2, 5).contiguous()   # [frag][ks][plane][lq][lr][e]
    xop = img.reshape(-1).to(dev)
    ws = pips_mixer_x3_stream(w["w1"], w["w2"], slices).to(dev)
    b1 = w["b1"].to(dev)
    part = torch.full((slices, R, 512), float("nan"), device=dev)
    ok(lib.sampt_pips_mix_mlp_x3(P(xop), P(ws), P(b1), P(part), nseq, slices, S()), "mix_mlp_x3")
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()
    scale = ref.abs().max().item()
    assert (part.cpu().double() - ref).abs().max().item() < 6e-6 * max(scale, 1.0)
    assert rel_err(part.sum(0), ref.sum(0)) < 4e-6


def test_pips_mixer_x3_equals_f32_paths(lib, dev):
    """A clip's chained windows through the split-fp16 mixer (sampt_pips_set_mixer(2, .)) and the two exact-f32 paths: same
    trajectories to fp32 round-off, identical in index space, same visibilities."""
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.weights import init_pips_state_dict
    from tests.util import disc_queries, synthetic_clip
    frames, centres = synthetic_clip(T=12, H=128, W=256, seed=5)
    q = disc_queries(centres, n_pos=5, r=9.0)[None]
    outs = []
    try:
        for mode, wgs in ((0, 32), (2, 16), (2, 32)):
            ok(lib.sampt_pips_set_mixer(mode, wgs), "set_mixer")
            trk = PipsPointTracker(state_dict=init_pips_state_dict(72)).to(dev)
            tr, vi = trk(frames.to(dev)[None], q.to(dev))
            outs.append((tr.cpu(), vi.cpu()))
    finally:
        ok(lib.sampt_pips_set_mixer(2, 16), "set_mixer")       # the library's default
    for o in outs[1:]:
        print(f"\n[mixer x3 vs f32] max |d traj| = {max_abs(outs[0][0], o[0]):.3e} px")
        assert max_abs(outs[0][0], o[0]) < 2e-3, "trajectories of the two mixer paths differ"
        assert (outs[0][0].round() == o[0].round()).all()
        assert (outs[0][1] == o[1]).all()
