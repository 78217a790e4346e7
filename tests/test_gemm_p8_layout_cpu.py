"""CPU model of the W half-tile LDS layout of the ViT GEMM kernel (sam_pt_amd/csrc/gemm_f16_p8.hip): the kernel's index arithmetic
restated in numpy for the shipped fragment-to-column mapping (``ilv=False``) and for the interleaved mapping that was measured and
dropped in round 4 (``ilv=True``: 8 consecutive columns per lane for 16-byte fp16 stores; see the kernel's header), checked for
(1) correctness of the gather — the lane (lr, lq) of W fragment f receives the 16-byte k-chunk (kk*4 + lq) of exactly the W row the
mapping assigns it, given where the LDS-DMA stage put each chunk — (2) the column a lane's accumulator registers then hold in the
epilogue (8 consecutive columns per lane and half with the interleaved mapping), and (3) bank-conflict freedom of the fragment
reads under the bank model of /opt/skills/guides/cdna_hip_programming.md (bank = (byte / 4) % 64, ds_read_b128 serviced in 16-lane
groups of which 8 lanes share a clock).  Measured counterpart: SQ_LDS_BANK_CONFLICT in profiles/r*_gemm_sq_counters.txt."""
import pytest


def _stage_image(ilv):
    """LDS chunk (16 B) index -> (tile row 0..127, source k-chunk 0..7) as the B-slot stage writes it: wave w, instruction i,
    lane l lands at byte (i*8 + w)*1024 + l*16, i.e. tile row (i*8 + w)*8 + (l >> 3), chunk slot l & 7."""
    img = {}
    for w in range(8):
        for i in range(2):
            for lane in range(64):
                sub = lane >> 3
                row = (i * 8 + w) * 8 + sub                       # half-tile row = wn*32 + local, local = (w & 3)*8 + sub
                key = ((sub & 3) | ((w & 1) << 2)) if ilv else sub
                img[row * 8 + (lane & 7)] = (row, (lane & 7) ^ key)
    assert len(img) == 128 * 8
    return img


def _read(ilv, wn, f, kk, lane):
    lr, lq = lane & 15, lane >> 4
    b_rd = (wn * 32 + ((lr >> 2) * 8 + (lr & 3) if ilv else lr)) * 128 + ((lq ^ (lr & 7)) << 4)
    frag = (4 if ilv else 16) * 128
    return f * frag + (b_rd ^ (64 if kk else 0))


@pytest.mark.parametrize("ilv", [False, True])
def test_w_fragment_gather_and_columns(ilv):
    img = _stage_image(ilv)
    for wn in range(4):
        cols_of_lane = {}
        for f in range(2):
            for kk in range(2):
                for lane in range(64):
                    lr, lq = lane & 15, lane >> 4
                    byte = _read(ilv, wn, f, kk, lane)
                    assert byte % 16 == 0
                    row, chunk = img[byte // 16]
                    local = ((lr >> 2) * 8 + f * 4 + (lr & 3)) if ilv else (f * 16 + lr)
                    assert row == wn * 32 + local and chunk == kk * 4 + lq, (ilv, wn, f, kk, lane)
            # epilogue: output lane (lr', lq') register r of this fragment is W row n = lq'*4 + r of the fragment
            for lq in range(4):
                for r in range(4):
                    n = lq * 4 + r
                    local = ((n >> 2) * 8 + f * 4 + (n & 3)) if ilv else (f * 16 + n)
                    cols_of_lane.setdefault(lq, []).append(local)
        for lq, cols in cols_of_lane.items():
            if ilv:
                assert sorted(cols) == list(range(lq * 8, lq * 8 + 8))            # 8 consecutive columns: one 16-byte fp16 store
            else:
                assert sorted(cols) == list(range(lq * 4, lq * 4 + 4)) + list(range(16 + lq * 4, 20 + lq * 4))


@pytest.mark.parametrize("ilv", [False, True])
def test_w_fragment_reads_are_bank_conflict_free(ilv):
    for wn in range(4):
        for f in range(2):
            for kk in range(2):
                for lq in range(4):                      # a 16-lane group = the lanes of one lq
                    for half in range(2):                # 8 lanes x 16 B = 128 B = 32 banks per clock; two clocks per group
                        banks = []
                        for lr in range(half * 8, half * 8 + 8):
                            byte = _read(ilv, wn, f, kk, lq * 16 + lr)
                            banks += [((byte + o) // 4) % 64 for o in range(0, 16, 4)]
                        assert len(set(banks)) == 32, (ilv, wn, f, kk, lq, half)
