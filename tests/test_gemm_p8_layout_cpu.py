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


# ------------------------------------------------------------------------------------------------------------------------
# The stage / read / wait ORDER of the kernel, both schedules (template parameter SR; round 5 moved the LDS-DMA of a phase from its
# multiply segment into its read segment and the rotation on by one phase).  The kernel's header argues RAW and WAR by counting
# barriers; this is that argument as a program.  Model: a workgroup's waves form two groups, leaders (wm = 0) and laggers (wm = 1,
# one barrier behind).  Barrier instances are numbered I = 0, 1, 2, ...; for a leader, phase p (p = 4 * K-tile + phase) has its
# read segment in the interval (2p, 2p + 1) — between instances 2p and 2p + 1 — and its multiply segment in (2p + 1, 2p + 2); a
# lagger's are one instance later.  A ds_read issued in a read segment has returned by the END of the same phase's multiply segment
# (its MFMAs consume it).  An LDS-DMA becomes visible to a wave only after the ISSUING wave's vmcnt wait covered it AND a barrier
# instance that lies after that wait in the issuer and before the read in the reader.
SLOTS = ("A0", "A1", "B0", "B1")
READS = {0: ("B0", "A0"), 1: ("B1",), 2: ("A1",), 3: ("B0",)}            # ds_reads of phase P1..P4 (buffer = K-tile parity)


def _stage_of(sr, t, ph):
    """(buffer, slot, K-tile staged, 'read' | 'mult' segment) of the stage issued in phase ph of K-tile t."""
    cur = t % 2
    if sr:
        return {0: (cur ^ 1, "A1", t + 1), 1: (cur ^ 1, "B0", t + 1), 2: (cur, "A0", t + 2), 3: (cur, "B1", t + 2)}[ph] + ("read",)
    return {0: (cur ^ 1, "B0", t + 1), 1: (cur, "A0", t + 2), 2: (cur, "B1", t + 2), 3: (cur, "A1", t + 2)}[ph] + ("mult",)


def _segment(group, p, kind):
    """Barrier interval (lo, hi) of the read / multiply segment of phase p for leaders (group 0) / laggers (1)."""
    lo = 2 * p + (1 if kind == "mult" else 0) + group
    return lo, lo + 1


@pytest.mark.parametrize("sr", [True, False])
def test_stage_schedule_raw_and_war_by_barrier_count(sr):
    NT = 8                                                   # K-tiles simulated (the pattern repeats every two)
    # what the prologue leaves: K-tile 0 complete (waited, behind the prologue barrier = instance 0), and of K-tile 1 the half tiles
    # the in-loop rotation no longer stages (SR: A0, B1; round-3 schedule: A0, B1, A1) in flight
    staged = {}                                              # (buffer, slot) -> list of (K-tile, issue interval per group, waited-by instance per group)
    for slot in SLOTS:
        staged[(0, slot)] = [(0, {0: (-1, 0), 1: (-1, 0)}, {0: 0, 1: 0})]
    pro = ("A0", "B1") if sr else ("A0", "B1", "A1")
    for slot in SLOTS:
        staged[(1, slot)] = [(1, {0: (-1, 0), 1: (-1, 0)}, None)] if slot in pro else []
    issue_log = []                                           # stages in issue order per group: (key, index into staged[key])
    for slot in pro:
        issue_log.append(((1, slot), 0))
    reads_done = {}                                          # (buffer, slot) -> latest instance by which every read so far has returned
    for t in range(NT):
        for ph in range(4):
            p = 4 * t + ph
            # ---- reads of this phase: RAW
            for slot in READS[ph]:
                key = (t % 2, slot)
                ktile, _, waited = staged[key][-1]
                assert ktile == t, (sr, t, ph, slot, "holds K-tile", ktile)
                assert waited is not None, (sr, t, ph, slot, "read before any vmcnt wait covered its stage")
                for g in (0, 1):
                    rd_lo, _ = _segment(g, p, "read")
                    # every issuing group's wait lies before an instance that the reader has passed when its read segment starts
                    assert all(waited[gi] <= rd_lo for gi in (0, 1)), (sr, t, ph, slot, g, waited, rd_lo)
                    reads_done[key] = max(reads_done.get(key, 0), _segment(g, p, "mult")[1])
            # ---- the stage of this phase: WAR against every read of the slot's previous content
            buf, slot, kt, seg = _stage_of(sr, t, ph)
            key = (buf, slot)
            iss = {g: _segment(g, p, seg) for g in (0, 1)}
            for g in (0, 1):
                assert reads_done.get(key, 0) <= iss[g][0], (sr, t, ph, "stage of", key, "issued in", iss[g], "reads return by", reads_done.get(key))
            staged[key].append((kt, iss, None))
            issue_log.append((key, len(staged[key]) - 1))
            # ---- P4's vmcnt(4): everything but the two youngest stages (2 instructions each) has landed in the issuing wave
            if ph == 3:
                # SR: the wait follows P4's own stage in the read segment; round-3 schedule: it precedes P4's stage (multiply segment)
                upto = len(issue_log) - 2 if sr else len(issue_log) - 1 - 2
                for key2, idx in issue_log[:upto]:
                    kt2, iss2, waited2 = staged[key2][idx]
                    if waited2 is None:
                        # the wait sits in the read segment of this phase: a reader sees the data after the NEXT instance of that group
                        staged[key2][idx] = (kt2, iss2, {g: _segment(g, p, "read")[1] for g in (0, 1)})
    # every K-tile's four half tiles were staged exactly once, into the buffer of its parity
    for (buf, slot), lst in staged.items():
        kts = [k for k, _, _ in lst]
        assert kts == sorted(set(kts)) and all(k % 2 == buf for k in kts), (buf, slot, kts)


def test_stage_schedule_model_rejects_the_naive_move(monkeypatch):
    """Why round 5 also advanced the rotation: the round-3 rotation (a half tile re-staged ONE phase after its last read) with the
    stage merely moved into the read segment is a WAR hazard — a lagger's reads of the half tile have not returned by barrier count —
    and the model above says so."""
    import sys
    mod = sys.modules[__name__]

    def naive(sr, t, ph):
        cur = t % 2
        return {0: (cur ^ 1, "B0", t + 1), 1: (cur, "A0", t + 2), 2: (cur, "B1", t + 2), 3: (cur, "A1", t + 2)}[ph] + ("read",)
    monkeypatch.setattr(mod, "_stage_of", naive)
    with pytest.raises(AssertionError):
        test_stage_schedule_raw_and_war_by_barrier_count(False)
