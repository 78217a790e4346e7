"""CPU model of k_pips_mix_mlp's data flow (sam_pt_amd/csrc/pips_mixer.hip): a lane-level emulation of
v_mfma_f32_16x16x4_f32 and the kernel's operand / fragment index algebra, checked against the plain matrix products.

What it pins (the part a compile cannot): (i) with the weights as the first operand and the activations as the second, the
accumulator of lane (lr, lq) holds row lr and hidden units 4 lq .. 4 lq + 3 — which is exactly the second product's operand
layout, so the hidden activations never need a shuffle; (ii) the k index a lane supplies to MFMA j of a 16-deep chunk is the same
for both operands; (iii) the (slice, wave) -> hidden range, row group -> rows and the slab addressing; (iv) the LDS reduction's
owner mapping (output fragment g * 4 + wave is finished by `wave`)."""
import numpy as np
import pytest


def mfma_16x16x4(a, b, acc):
    """One wave-wide v_mfma_f32_16x16x4_f32: a, b [64] (one f32 per lane), acc [64][4].  Lane l supplies A[i = l & 15][k = l >> 4]
    and B[k = l >> 4][j = l & 15]; lane l, register r receives D[i = 4 (l >> 4) + r][j = l & 15]
    (cdna_hip_programming.md section 3; the epilogue of csrc/gemm.hip gemm_kernel relies on the same map)."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a[l]
        B[l >> 4, l & 15] = b[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[4 * (l >> 4) + r, l & 15]
    return out


@pytest.mark.parametrize("NS,nseq", [(8, 3), (16, 2)])
def test_mix_mlp_index_algebra(NS, nseq):
    rng = np.random.default_rng(NS + nseq)
    MD, MH = 512, 2048
    R = nseq * 8
    NF = MH // (NS * 4) // 16
    xn = rng.standard_normal((R, MD))                     # the LayerNorm'ed rows (the model starts after the normalisation)
    w1 = rng.standard_normal((MH, MD)) / 20
    b1 = rng.standard_normal(MH) / 10
    w2 = rng.standard_normal((MD, MH)) / 40
    act = np.tanh                                         # any elementwise function: the layout is what is under test
    h_ref = act(xn @ w1.T + b1)
    rgs = (nseq + 1) // 2
    part = np.full((NS, R, MD), np.nan)
    lanes = np.arange(64)
    lr, lq = lanes & 15, lanes >> 4
    # a reduced model: one (row group, slice) pair per configuration is enough to pin the maps, all four waves of it
    for block in (0, rgs * NS - 1, (rgs * NS) // 2):
        sl, rg = block % NS, block // NS
        r0 = rg * 16
        row = np.minimum(r0 + lr, R - 1)
        red = np.zeros((4, MD // 16, 64, 4))
        for wave in range(4):
            h0 = (sl * 4 + wave) * 16 * NF
            acc1 = [np.zeros((64, 4)) for _ in range(NF)]
            for c in range(32):
                xa = np.stack([xn[row, c * 16 + lq * 4 + j] for j in range(4)], 1)             # float4 of the lane
                for f in range(NF):
                    wv = np.stack([w1[h0 + f * 16 + lr, c * 16 + lq * 4 + j] for j in range(4)], 1)
                    for j in range(4):
                        acc1[f] = mfma_16x16x4(wv[:, j], xa[:, j], acc1[f])
            gh = []
            for f in range(NF):
                bq = np.stack([b1[h0 + f * 16 + lq * 4 + r] for r in range(4)], 1)
                gh.append(act(acc1[f] + bq))
                # (i): lane (lr, lq), register r == hidden unit h0 + 16 f + 4 lq + r of row lr
                for r in range(4):
                    np.testing.assert_allclose(gh[f][:, r], h_ref[row, h0 + f * 16 + lq * 4 + r], rtol=1e-10, atol=1e-12)
            for o in range(MD // 16):
                acc2 = np.zeros((64, 4))
                for f in range(NF):
                    wv = np.stack([w2[o * 16 + lr, h0 + f * 16 + lq * 4 + j] for j in range(4)], 1)
                    for j in range(4):
                        acc2 = mfma_16x16x4(wv[:, j], gh[f][:, j], acc2)
                red[wave, o] = acc2
        for g in range(MD // 16 // 4):
            for wave in range(4):                                      # the wave that finishes output fragment g * 4 + wave
                o = g * 4 + wave
                v = red[0, o] + red[1, o] + red[2, o] + red[3, o]
                for l in range(64):
                    if r0 + lr[l] < R:
                        part[sl, r0 + lr[l], o * 16 + lq[l] * 4: o * 16 + lq[l] * 4 + 4] = v[l]
        hs = MH // NS
        ref = h_ref[:, sl * hs:(sl + 1) * hs] @ w2[:, sl * hs:(sl + 1) * hs].T
        rows = slice(r0, min(r0 + 16, R))
        np.testing.assert_allclose(part[sl, rows], ref[rows], rtol=1e-9, atol=1e-10)


def test_mix_slice_selection_and_grid():
    """pips_mix_slices' rule restated: 8 / 16 / 32 hidden slices so that row groups x slices reaches the workgroup target."""
    def slices(nseq, target=32):
        rgs = (nseq + 1) // 2
        return 8 if rgs * 8 >= target else (16 if rgs * 16 >= target else 32)
    assert [slices(n) for n in (1, 2, 3, 4, 5, 8, 24)] == [32, 32, 16, 16, 16, 8, 8]
    assert slices(8, 64) == 16 and slices(8, 128) == 32
    for n in (1, 4, 8, 24):                         # same-slice workgroups share blockIdx % 8 (one XCD's L2 holds the slice once)
        NS = slices(n)
        for b in range(((n + 1) // 2) * NS):
            assert (b % NS) % 8 == b % 8


# ---- split-fp16 generation (csrc/pips_mixer_x3.hip): packed weight stream + operand images + v_mfma_f32_16x16x32_f16 ----------
def mfma_16x16x32(a, b, acc):
    """a, b [64][8] (8 halves per lane, as float64 here), acc [64][4].  Lane l supplies A[i = l & 15][k = 8 (l >> 4) + e] and
    B[k = 8 (l >> 4) + e][j = l & 15]; lane l, register r receives D[4 (l >> 4) + r][l & 15] (csrc/gemm.hip's fp16 path)."""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4): 8 * (l >> 4) + 8] = a[l]
        B[8 * (l >> 4): 8 * (l >> 4) + 8, l & 15] = b[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[4 * (l >> 4) + r, l & 15]
    return out


@pytest.mark.parametrize("NS", [16, 32])
def test_mix_mlp_x3_stream_and_operand_images(NS):
    """The kernel's consumption of pack.pips_mixer_x3_stream replayed on the CPU: stage / image indices, the k-slot permutation
    that lets two accumulator fragments of the first product be one 32-deep operand of the second, the operand-image layout
    k_pips_mix_pre writes, the 2^8 / 2^6 / 2^-14 scaling and the three-term product's accuracy."""
    import torch
    from sam_pt_amd.pack import MIXER_X3_ASHIFT, pips_mixer_x3_stream
    g = torch.Generator().manual_seed(NS)
    w1 = torch.randn(2048, 512, generator=g) / 22
    w2 = torch.randn(512, 2048, generator=g) / 45
    b1 = torch.randn(2048, generator=g) / 10
    y = torch.randn(16, 512, generator=g) * 1.3                     # one row fragment of LayerNorm2 outputs
    NF = 2048 // NS // 16
    stream = pips_mixer_x3_stream(w1, w2, NS)
    assert stream.shape == (NS, 64 * NF * 512) and stream.dtype == torch.float16
    # operand images as k_pips_mix_pre writes them: [ks][plane][lane][8], lane (lr, lq): row lr, k = 32 ks + 8 lq + e
    ys = y * float(1 << MIXER_X3_ASHIFT)
    yhi = ys.half()
    ylo = (ys - yhi.float()).half()
    xop = torch.zeros(16, 2, 64, 8, dtype=torch.float16)
    for row in range(16):
        for c4 in range(0, 512, 4):
            ks, lq, e0 = c4 >> 5, (c4 & 31) >> 3, c4 & 7
            xop[ks, 0, row + 16 * lq, e0:e0 + 4] = yhi[row, c4:c4 + 4]
            xop[ks, 1, row + 16 * lq, e0:e0 + 4] = ylo[row, c4:c4 + 4]
    sl = NS // 3
    h0 = sl * 16 * NF
    imgs = stream[sl].view(-1, 64, 8).double().numpy()              # the linear stream as 1-KB images
    xo = xop.double().numpy()
    KS_PER, O_PER = 16 // NF, 32 // NF
    acc1 = [np.zeros((64, 4)) for _ in range(NF)]
    hh = None
    part = np.zeros((16, 512))
    lanes = np.arange(64)
    lr, lq = lanes & 15, lanes >> 4
    act = np.tanh
    for t in range(2 * NF):
        st = imgs[32 * t: 32 * t + 32]
        if t < NF:
            for kk in range(KS_PER):
                ks = t * KS_PER + kk
                for f in range(NF):
                    img = (kk * NF + f) * 2
                    whi, wlo = st[img], st[img + 1]
                    acc1[f] = mfma_16x16x32(wlo, xo[ks, 0], acc1[f])
                    acc1[f] = mfma_16x16x32(whi, xo[ks, 1], acc1[f])
                    acc1[f] = mfma_16x16x32(whi, xo[ks, 0], acc1[f])
            if t == NF - 1:
                pre = y.double().numpy() @ w1.double().numpy()[h0:h0 + 16 * NF].T       # (16, 16 NF)
                hid = np.zeros((16, 16 * NF))
                hh = np.zeros((NF // 2, 2, 64, 8))
                for kp in range(NF // 2):
                    for e in range(8):
                        f, r = 2 * kp + (e >> 2), e & 3
                        unit = 16 * f + 4 * lq + r
                        v = acc1[f][:, r] / 16384.0
                        np.testing.assert_allclose(v, pre[lr, unit], rtol=0, atol=3e-6 * np.abs(pre).max())
                        gg = act(v + b1.numpy()[h0 + unit])
                        hid[lr, unit] = gg
                        s = torch.from_numpy(gg * 64.0).float()
                        hi = s.half()
                        hh[kp, 0, :, e] = hi.double().numpy()
                        hh[kp, 1, :, e] = (s - hi.float()).half().double().numpy()
        else:
            for oo in range(O_PER):
                o = (t - NF) * O_PER + oo
                a = np.zeros((64, 4))
                for kp in range(NF // 2):
                    img = (oo * (NF // 2) + kp) * 2
                    a = mfma_16x16x32(st[img + 1], hh[kp, 0], a)
                    a = mfma_16x16x32(st[img], hh[kp, 1], a)
                    a = mfma_16x16x32(st[img], hh[kp, 0], a)
                for l in range(64):
                    part[lr[l], 16 * o + 4 * lq[l]: 16 * o + 4 * lq[l] + 4] = a[l] / 16384.0
    ref = hid @ w2.double().numpy()[:, h0:h0 + 16 * NF].T
    np.testing.assert_allclose(part, ref, rtol=0, atol=3e-6 * np.abs(ref).max())
