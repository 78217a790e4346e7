"""CPU model of the LDS layouts of the fused ViT attention kernel (sam_pt_amd/csrc/attention.hip, k_flash_f16): the kernel's
index arithmetic restated in numpy and checked for (1) correctness of the gather — every lane of the PV step's transposing
read (ds_read_b64_tr_b16) and of the QK^T step's 16-byte read receives exactly the (key slot, channel) elements its MFMA
operand needs, given what the LDS-DMA wrote where — and (2) bank-conflict freedom under the bank model of
/opt/skills/guides/cdna_hip_programming.md §LDS (bank = (byte / 4) % 64; ds_read_b128 is serviced in 16-lane groups,
ds_read_b64_tr_b16 in 32-lane halves).  The constants below mirror the kernel (KSWZ / KSH, VP, the g_flash_pad page, vl[]);
the lane <-> address mapping of the transposing read is the one tools/probes/tr_probe.hip printed on an MI355X
(profiles/r3_tr_probe.txt).  Measured counterpart: profiles/r3_attn_sq_counters.txt (SQ_LDS_BANK_CONFLICT 0.8 - 2.4 % of
SQ_LDS_IDX_ACTIVE)."""
import numpy as np
import pytest

PAD = ("pad", 0)   # marker for the 16-byte chunks that come from g_flash_pad


def _geometry(HD):
    DT = (HD + 31) // 32
    lrow = DT * 32 > HD
    VP = DT * 32 if lrow else HD
    CPR, CPV = HD // 8, VP // 8
    KSWZ = 7 if CPR == 8 else (3 if CPR == 4 else (1 if CPR == 10 else 0))
    KSH = 2 if CPR == 4 else (3 if CPR == 10 else 1)
    return DT, lrow, VP, CPR, CPV, KSWZ, KSH


def _dma_images(HD):
    """What dma_tile leaves in LDS: per 16-byte LDS chunk, the (slot, source chunk) it holds.  Instruction i, lane l writes
    LDS bytes [i*1024 + l*16, +16)."""
    DT, lrow, VP, CPR, CPV, KSWZ, KSH = _geometry(HD)
    kimg, vimg = {}, {}
    for i in range(CPR):
        for lane in range(64):
            e = i * 64 + lane
            slot, c = divmod(e, CPR)
            kimg[e] = (slot, c ^ ((slot >> KSH) & KSWZ))
    for i in range(CPV):
        for lane in range(64):
            e = i * 64 + lane
            slot, c = divmod(e, CPV)
            cs = c ^ (((slot >> 1) & 1) << 2) if HD == 64 else c
            vimg[e] = (slot, cs) if c < CPR else ("pad", c - CPR)
    assert len(kimg) == 64 * CPR and len(vimg) == 64 * CPV          # every chunk of both tiles written exactly once
    return kimg, vimg


def _banks(byte, nbytes):
    return {((byte + o) // 4) % 64 for o in range(0, nbytes, 4)}


@pytest.mark.parametrize("HD", [80, 64, 32])
def test_k_fragment_reads(HD):
    """QK^T: lane (li, hi) of k-step ks reads 16 bytes = channels ks*16 + hi*8 .. +7 of key slot kt*32 + li."""
    DT, lrow, VP, CPR, CPV, KSWZ, KSH = _geometry(HD)
    kimg, _ = _dma_images(HD)
    for kt in range(2):
        for ks in range(HD // 16):
            addr = {}
            for lane in range(64):
                li, hi = lane & 31, lane >> 5
                krow = kt * 32 + li
                chunk = (ks * 2 + hi) ^ ((krow >> KSH) & KSWZ)
                byte = (krow * HD + chunk * 8) * 2                   # Ks[buf][krow][chunk * 8]
                assert kimg[byte // 16] == (krow, ks * 2 + hi)       # the chunk the DMA put there is the one the MFMA needs
                addr[lane] = byte
            for g in range(4):                                       # ds_read_b128: four 16-lane groups, one cycle each
                seen = {}
                for lane in range(16 * g, 16 * g + 16):
                    for b in _banks(addr[lane], 16):
                        assert seen.setdefault(b, addr[lane]) == addr[lane], f"HD {HD}: K bank conflict in group {g}"


@pytest.mark.parametrize("HD", [80, 64, 32])
def test_v_transposing_reads(HD):
    """PV: for k-step t, quad q (keys 16t + 8q + 4hi + {0..3}) and d-tile dt, lane (li, hi) must receive channel dt*32 + li
    of those four key slots.  Hardware (tr_probe): in a 16-lane group, source lane s supplies 8 bytes = 4 consecutive
    channels of ONE key; result lane c, element j = element (c & 3) of source lane (c >> 2) + 4 j."""
    DT, lrow, VP, CPR, CPV, KSWZ, KSH = _geometry(HD)
    _, vimg = _dma_images(HD)
    pad_page = [1.0] + [0.0] * 15

    def lds_half(byte):                                              # what the DMA left at this 2-byte LDS position
        what, cs = vimg[byte // 16]
        idx = (byte % 16) // 2
        if what == "pad":
            return ("const", pad_page[cs * 8 + idx])
        return (what, cs * 8 + idx)                                  # (key slot, channel)

    for dt in range(DT):
        for t in range(4):
            for q in range(2):
                addr = {}
                for lane in range(64):
                    li, hi = lane & 31, lane >> 5
                    s16, g16 = li & 15, li >> 4
                    r4, cc = s16 >> 2, s16 & 3
                    vl = (4 * hi + r4) * VP * 2 + ((dt * 64 + g16 * 32 + cc * 8) ^ ((((r4 >> 1) & 1) << 6) if HD == 64 else 0))
                    addr[lane] = vl + (16 * t + 8 * q) * VP * 2
                    assert addr[lane] % 8 == 0 and addr[lane] + 8 <= 64 * VP * 2
                for lane in range(64):
                    li, hi = lane & 31, lane >> 5
                    base, c = lane & ~15, lane & 15
                    for j in range(4):
                        src = base + (c >> 2) + 4 * j                 # the lane whose 8 bytes element j comes from
                        got = lds_half(addr[src] + 2 * (c & 3))
                        d = dt * 32 + li
                        if d < HD:
                            assert got == (16 * t + 8 * q + 4 * hi + j, d), (HD, dt, t, q, lane, j, got)
                        else:                                        # padded channels: channel HD is all ones, the rest zero
                            assert got == ("const", 1.0 if d == HD else 0.0), (HD, dt, t, q, lane, j, got)
                for half in range(2):                                # serviced in two 32-lane halves
                    seen = {}
                    for lane in range(32 * half, 32 * half + 32):
                        for b in _banks(addr[lane], 8):
                            assert seen.setdefault(b, addr[lane]) == addr[lane], f"HD {HD}: V bank conflict (dt {dt}, t {t})"


def test_p_operand_slot_order_matches_the_v_quads():
    """The kernel packs P as pb[kt * 2 + (r >> 3)][r & 7] = exp2(score register r of S^T tile kt), and register r of lane-half
    hi of a 32x32 C tile is row (r & 3) + 8 (r >> 2) + 4 hi.  So element j of the B operand of PV step t is key slot
    16t + 4hi + (j & 3) + 8 (j >> 2): elements 0..3 are the first transposing read's four keys, 4..7 the second's."""
    for kt in range(2):
        for r in range(16):
            for hi in range(2):
                t, j = kt * 2 + (r >> 3), r & 7
                p_slot = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi
                v_slot = 16 * t + 8 * (j >> 2) + 4 * hi + (j & 3)     # quad j >> 2, key j & 3 of test_v_transposing_reads
                assert p_slot == v_slot


# ---------------------------------------------------------------------------------------------------------------------------
# The per-workgroup DMA address table (dma_tab) and the XCD-aware work order (common.h flash_wg_decode), restated.
# ---------------------------------------------------------------------------------------------------------------------------
def _tile_geometry(SG):
    RPT = 1 if SG >= 64 else 64 // SG
    KTV = 64 if SG >= 64 else RPT * SG
    return RPT, KTV


def _dma_table(HD, NW, SG, wave, px0, gw, padded):
    """dma_tab rows of one wave: per instruction and lane the packed word the kernel builds once per workgroup."""
    DT, lrow, VP, CPR, CPV, KSWZ, KSH = _geometry(HD)
    RPT, KTV = _tile_geometry(SG)
    rows = []
    for cp, is_v in ((CPR, False), (CPV, True)):
        for i in range(wave, cp, NW):
            words = []
            for lane in range(64):
                slot, c = divmod(i * 64 + lane, cp)
                sc = min(slot, KTV - 1)
                siy, six = divmod(sc, SG)
                if not is_v:
                    off, page = (c ^ ((slot >> KSH) & KSWZ)) * 16, 0
                else:
                    cs = c ^ (((slot >> 1) & 1) << 2) if HD == 64 else c
                    off, page = (cs * 16, 0) if c < CPR else ((c - CPR) * 16, 1)
                colpad = 1 if (padded and px0 + six >= gw) else 0
                words.append(sc | siy << 8 | colpad << 12 | page << 13 | off << 16)
            rows.append((is_v, i, words))
    return rows


@pytest.mark.parametrize("HD,NW,SG", [(80, 4, 64), (64, 4, 64), (80, 4, 14), (64, 4, 14), (32, 4, 16), (32, 2, 6)])
def test_dma_table_reproduces_the_dma_image(HD, NW, SG):
    """Decoding the table gives, for every LDS chunk, the (slot, source chunk) of the direct formula (_dma_images) — the fields
    fit their bit ranges, every instruction of the tile is issued by exactly one wave."""
    DT, lrow, VP, CPR, CPV, KSWZ, KSH = _geometry(HD)
    RPT, KTV = _tile_geometry(SG)
    kimg, vimg = _dma_images(HD)
    seen_k, seen_v = set(), set()
    for wave in range(NW):
        for is_v, i, words in _dma_table(HD, NW, SG, wave, 0, 0, False):
            for lane, t in enumerate(words):
                e = i * 64 + lane
                sc, siy, page, off = t & 255, (t >> 8) & 15, (t >> 13) & 1, t >> 16
                assert t < 2 ** 32 and siy == sc // SG and siy < max(RPT, 1)
                assert sc == min(e // (CPV if is_v else CPR), KTV - 1) and off % 16 == 0
                if is_v and page:
                    assert vimg[e] == ("pad", off // 16)
                else:
                    assert (vimg if is_v else kimg)[e] == (e // (CPV if is_v else CPR), off // 16)
                (seen_v if is_v else seen_k).add(e)
    assert len(seen_k) == 64 * CPR and len(seen_v) == 64 * CPV


@pytest.mark.parametrize("SG,nwx,nwy,gh,gw", [(14, 3, 2, 22, 36), (14, 5, 5, 64, 64), (6, 3, 3, 16, 16), (14, 5, 3, 42, 64)])
def test_dma_sources_never_touch_padded_rows(SG, nwx, nwy, gh, gw):
    """FlashPad: whatever the slot — a key, a pad slot of the tile, a grid row beyond the window — the row a DMA lane fetches is
    either the bias row or the qkv row of a REAL token of the same window, and a real key always fetches its own row (the
    padded rows of the qkv matrix are never written by the qkv GEMM)."""
    HD, NW = 80 if SG == 14 else 32, 4 if SG == 14 else 2
    RPT, KTV = _tile_geometry(SG)
    N = SG * SG
    for w in range(nwx * nwy):
        py0, px0 = (w // nwx) * SG, (w % nwx) * SG
        for wave in range(NW):
            for is_v, i, words in _dma_table(HD, NW, SG, wave, px0, gw, True):
                for kt0, kh0 in zip(range(0, N, KTV), range(0, SG + RPT, RPT)):
                    for t in words:
                        sc, siy, colpad = t & 255, (t >> 8) & 15, (t >> 12) & 1
                        if (t >> 13) & 1:
                            continue                                        # constant pad page
                        krow = min(kt0 + sc, N - 1)
                        gr = kh0 + siy
                        from_bias = bool(colpad) or gr >= SG or py0 + gr >= gh
                        iy, ix = divmod(kt0 + sc, SG)
                        real_key = kt0 + sc < N and py0 + iy < gh and px0 + ix < gw
                        if real_key:
                            assert not from_bias and krow == kt0 + sc
                        if not from_bias:                                   # the row read is a real token of this window
                            ry, rx = divmod(krow, SG)
                            assert py0 + ry < gh and px0 + rx < gw


def _wg_decode(L, nx, heads, B, uh):
    if uh <= 0:
        return L % nx, (L // nx) % heads, L // nx // heads
    hg = heads // uh
    units, pu = B * hg, uh * nx
    full = units & ~7
    if L < full * pu:
        j = L >> 3
        unit, r = (j // pu) * 8 + (L & 7), j % pu
    else:
        unit, r = full + (L - full * pu) // pu, (L - full * pu) % pu
    return r % nx, (unit % hg) * uh + r // nx, unit // hg


@pytest.mark.parametrize("nx,heads,B,uh", [(2, 16, 200, 16), (2, 16, 19, 16), (2, 2, 3, 2), (32, 16, 8, 1), (32, 3, 3, 1),
                                           (32, 16, 8, 0), (1, 2, 11, 2)])
def test_flash_work_order_is_a_bijection_that_keeps_units_on_one_xcd(nx, heads, B, uh):
    """common.h flash_wg_decode: every (query block, head, b) exactly once; with uh > 0 the workgroups of a unit of the first
    floor(units / 8) * 8 units all have the same L & 7 (= the XCD of a 1-D grid's round-robin dispatch)."""
    total = nx * heads * B
    seen = {}
    for L in range(total):
        x, h, b = _wg_decode(L, nx, heads, B, uh)
        assert 0 <= x < nx and 0 <= h < heads and 0 <= b < B
        assert (x, h, b) not in seen
        seen[(x, h, b)] = L
    assert len(seen) == total
    if uh > 0:
        hg = heads // uh
        full = (B * hg) & ~7
        for unit in range(full):
            b, g = divmod(unit, hg)
            xcds = {seen[(x, g * uh + hh, b)] & 7 for x in range(nx) for hh in range(uh)}
            assert len(xcds) == 1
