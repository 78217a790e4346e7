// fp16 MFMA GEMM for the ViT image encoder: 256 x 256 x 64 tiles, 8 waves, 8-phase LDS-DMA pipeline, persistent workgroups.
//
//   C[M][N] = epi(A[M][K] . W[N][K]^T + bias) (+ residual)        A, W fp16 K-contiguous; fp32 accumulate
//
// One workgroup (512 threads = 8 waves as 2 (M) x 4 (N), each wave a 128 x 64 patch = 8 x 4 fragments of
// v_mfma_f32_16x16x32_f16, 128 accumulator registers) per CU walks a list of output tiles.  The operand stream never
// stops at a tile boundary: K-tiles of the NEXT output tile are already in flight while the accumulators of the current
// one are written out.
//
// LDS (128 KiB): two K-tile buffers x four 16-KiB "half tiles" (128 rows x 64 halfs):
//     A0 = A rows {wm*128 +      0..63 }   A1 = A rows {wm*128 + 64..127}          (wm = 0, 1)
//     B0 = W rows {wn*64  +      0..31 }   B1 = W rows {wn*64  + 32..63 }          (wn = 0..3)
// so that every wave reads the quadrant (ha, hb) of its own 128 x 64 patch from half tiles (A ha, B hb).  Half tiles go
// HBM -> LDS with two `global_load_lds_dwordx4` per wave (no VGPR staging).  The LDS image of one instruction is linear
// (M0 base + lane*16), so the bank swizzle is applied to the SOURCE address: lane l fetches, for tile row r, the 16-byte
// chunk (l&7) ^ (r&7); fragment reads apply the same XOR (conflict-free for ds_read_b128's 16-lane groups).
//
// A K-tile is four phases, one quadrant of 16 MFMAs each, in the order (a0,b0) (a0,b1) (a1,b1) (a1,b0):
//     phase   ds_read            last read of   stage issued (after the phase's first barrier)
//     P1      B0 (4) + A0 (8)    A0             B0 of K-tile +1  (other buffer; its last read was the previous P4)
//     P2      B1 (4)             B1             A0 of K-tile +2  (this buffer)
//     P3      A1 (8)             A1             B1 of K-tile +2
//     P4      B0 (4)             B0             A1 of K-tile +2
// i.e. 7 half tiles are staged ahead of the one being multiplied.  `s_waitcnt vmcnt(4)` in P4's read segment (two half
// tiles may stay in flight ACROSS the barriers) retires the whole next K-tile; it is read from the next phase on.  The two
// wave rows run staggered by one barrier: while wm = 0 multiplies, wm = 1 (the other wave of each SIMD) reads fragments.
//
// Ordering (what makes it correct, not just fast):
//   RAW  LDS-DMA data is visible to a ds_read only after the ISSUING wave's vmcnt wait and a barrier both pass: the wait
//        sits before P4's first barrier in every wave (wm = 1 executes it one barrier later), the first read of that data
//        is in the next phase's read segment of wm = 0 - one more barrier in between.
//   WAR  a half tile is re-staged one phase after its last ds_read: all reads of a phase feed that phase's MFMAs, so they
//        have returned before the phase's second barrier in every wave; the stage is issued after the NEXT phase's first
//        barrier (wm = 1's second barrier of the reading phase is at or before it).
//
// Tile order: strips of R row panels, column-major inside a strip; XCD x (= blockIdx.x & 7, each XCD has a private 4 MiB
// L2) owns the x-th eighth of that list and its workgroups take consecutive tiles, so the tiles in flight on one XCD form
// an R x (32 / R) patch that shares A panels and W tiles through that L2 while the panels of a strip stay resident.
//
// Tried and dropped (round 4, profiles/r4_c3_*, r4_c4_*): interleaving the two W fragments of a 32-column half (fragment fj = W
// rows (n >> 2) * 8 + fj * 4 + (n & 3)) so that a lane of a 16-bit output owns 8 consecutive columns — one 16-byte store instead
// of two 8-byte ones, 16 rows x 64 contiguous bytes per instruction.  Isolated and in steady state the fp16-output launches
// gained 4 - 6 % (qkv 928 -> 982, fc1 870 -> 906 TFLOP/s; the stores are 13 % of a qkv launch), but inside the encoder the very
// same build lost 2 % twice in an A / B of one call (in situ 885 -> 867 TFLOP/s, 113.3 -> 110.3 fps), so the plain mapping stays.
// Where the epilogue time of the f32 + residual launches goes (proj, M = 32768, steady state, profiles/r4_c3_p8_diag.log): 146.6
// us as shipped, 121.8 without the stores, 129.8 without the residual loads, 97.4 without both, 92.8 without any epilogue.  vmcnt
// retires loads and stores in issue order, so the wait for residual batch b + 1 also waits for the stores of batch b to be
// acknowledged, and the 64 MB all CUs write at the end of a round of tiles drain at memory bandwidth while only 1.75 K-tiles
// of the next tile are prefetched.  A two-pass epilogue (all residual loads folded into the accumulators in place, then all
// stores) removes those waits on paper but costs ~100 spilled VGPRs (the kernel sits at 250 of 256): not shipped.
// The in-place residual add as f32 atomics in the L2 (C += v: no residual load, no round trip, the same single rounding) is correct
// and 3.4 x slower — proj at M = 32768: 562 vs 164 us, 128 global_atomic_add_f32 per lane and tile against 32 + 32 16-byte
// accesses (profiles/r4_c23_*): the L2 retires ~0.1 T dword atomics per second.
//
// GemmP::x3 (template X3): the operands are "x3 rows" (common.h: every 64-half K-tile row is [hi(32) | lo(32)] of 32 real k)
// and a phase issues 24 MFMAs instead of 16 — hi.hi + hi.lo + lo.hi per fragment pair, fp32-grade products at a third of
// the fp16 rate — on the very same stage / barrier schedule (the K-tile images in LDS are byte-identical in size and
// layout, only the fragment pairing changes).  OUT = 2 writes C as x3 rows too (the next GEMM's A operand).
//
// Launcher conditions: K % 128 == 0 (an even number of K-tiles: tile boundaries fall on buffer 0), N % 256 == 0, M >= 256.
// Rows beyond M are clamped to the last valid row for the loads and never stored.
#include "common.h"

namespace sampt {

typedef __attribute__((address_space(3))) void lds_void_p8;
typedef const __attribute__((address_space(1))) void glb_void_p8;

int g_p8_sched = 0;         // 0: LDS-DMA issued in the read segments (round 5); 1: behind the first MFMAs (rounds 3 - 4), for A / B runs
int g_p8_trim = 1;          // sampt_gemm_set_trim: 1 = launch the fewest workgroups per XCD that keep the number of tile rounds
int g_p8_stagger = 0;       // experiment knob (sampt_gemm_set_stagger): phase groups of the persistent workgroups, 0 / 1 = off

namespace {
constexpr int P8_HALF = 128 * 64 * 2;      // bytes of a half tile
constexpr int P8_BUF = 4 * P8_HALF;        // bytes of a K-tile buffer: A0 | A1 | B0 | B1
constexpr int P8_SLOT_A0 = 0, P8_SLOT_A1 = 1, P8_SLOT_B0 = 2, P8_SLOT_B1 = 3;
template <int V> struct IC { static constexpr int value = V; };
#ifndef P8_EPI_ROWS
#define P8_EPI_ROWS 2     // fragment rows (16 matrix rows each) whose residual is loaded in one batch: 2 x 4 x 16 B per lane
#endif
}  // namespace

// ACT: ACT_NONE or ACT_GELU (compile time; other activations are left to the generic kernels).  OUT: 0 = f32, 1 = f16,
// 2 = f16 x3 rows.  X3: 3-term split-fp16 products.
// SR (round 5): the two LDS-DMA instructions of a phase are issued in the phase's READ segment (before its first barrier, while the
// other wave of the SIMD multiplies) instead of behind the first two MFMAs of its multiply segment.  An LDS-DMA instruction costs
// its wave 60 - 185 issue cycles (MI355X_MICROARCH.md, per-instruction constants) — far more than the 32 cycles of matrix work two
// queued MFMAs hold — so inside the multiply segment the matrix pipe ran dry behind every stage.  Issued one barrier earlier the
// WAR distance would shrink to zero barriers, so the rotation moves on by one phase as well: a half tile is now re-staged TWO
// phases after its last read (6 half tiles ahead of the multiply instead of 7):
//     phase   ds_read            stage issued in the read segment        its last read was in
//     P1      B0 (4) + A0 (8)    A1 of K-tile +1  (other buffer)         P3 of the previous K-tile
//     P2      B1 (4)             B0 of K-tile +1  (other buffer)         P4 of the previous K-tile
//     P3      A1 (8)             A0 of K-tile +2  (this buffer)          P1
//     P4      B0 (4)             B1 of K-tile +2  (this buffer)          P2
//   WAR  leaders (wm = 0) issue the stage of phase p between barrier instances I(2p-1) and I(2p); the laggers' (wm = 1) reads of
//        phase p-2 feed their multiply segment between I(2p-3) and I(2p-2), so they have returned before I(2p-2) in every wave.
//   RAW  unchanged: `s_waitcnt vmcnt(4)` in P4's read segment, now AFTER P4's own stage — the two youngest stages (A0, B1 of
//        K-tile +2) may stay in flight, everything of K-tile +1 has landed in this wave; the barrier(s) before the first read of
//        K-tile +1 publish the other waves' parts exactly as before.
template <int ACT, int OUT, bool X3, bool SR>
__global__ __launch_bounds__(512, 2) void gemm_f16_p8(GemmP p) {
  constexpr bool STAGGER = true;   // the two wave rows run one barrier apart (without: -1.7 %, profiles/r3_v1_*)
  __shared__ __attribute__((aligned(1024))) char lds[2 * P8_BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- this workgroup's tile list: global order = strips of R row panels, column-major inside a strip
  const int nt_m = (p.M + 255) >> 8, nt_n = p.N >> 8, R = p.xcd_swizzle;
  const int ntiles = nt_m * nt_n;
  const int xcd = blockIdx.x & 7, nwg = gridDim.x >> 3;
  const int g_end = (int)(((long)ntiles * (xcd + 1)) >> 3);
  int g_cmp = (int)(((long)ntiles * xcd) >> 3) + (int)(blockIdx.x >> 3);
  if (g_cmp >= g_end) return;
  // Phase groups (p.p8_stagger = G > 1): every tile of a launch takes the same time, so all workgroups reach their epilogues
  // together and the 33 - 115 MB a round of tiles writes (and, for the in-place residual, reads) arrive as one burst that drains
  // at the memory system's rate while no matrix pipe is busy.  Group g of G starts g / G of a tile's K-loop late: the bursts of
  // the groups interleave with the other groups' K-loops.  (~1500 shader cycles per K-tile: 20 K-tiles = 12 - 15 us.)
  if (p.p8_stagger > 1) {
    const int grp = (int)(blockIdx.x >> 3) % p.p8_stagger;
    if (grp) {
      const long long wait = (long long)grp * (p.K >> 6) * 1500 / p.p8_stagger, t0 = (long long)__builtin_amdgcn_s_memtime();
      while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
  }
  const int per_strip = R * nt_n, nfull = nt_m / R;
  auto decode = [&](int g, int& tm, int& tn) {
    int strip = g / per_strip, t, rows;
    if (strip < nfull) t = g - strip * per_strip, rows = R;
    else strip = nfull, t = g - nfull * per_strip, rows = nt_m - nfull * R;
    tn = t / rows;
    tm = strip * R + (t - tn * rows);
  };

  const char* __restrict__ A = (const char*)p.A;
  const char* __restrict__ W = (const char*)p.W;
  const int nk = p.K >> 6;

  // ---- stage cursor (runs 7 half tiles ahead of the multiply, across output tiles)
  const int sub = lane >> 3, chunk = (lane & 7) ^ sub;
  unsigned a_so[2][2];                    // [half][instruction]: byte offset of this lane's A row + swizzled chunk
  const unsigned b_so = (unsigned)(((wave >> 2) * 64 + (wave & 3) * 8 + sub) * p.ldw * 2 + chunk * 16);
  long b_tile = 0;                        // uniform: byte offset of the staged tile's first W row
  int s_kt = 0, g_stg = g_cmp;
  auto stage_tile = [&](int g) {
    int tm, tn;
    decode(g, tm, tn);
    b_tile = (long)(tn << 8) * p.ldw * 2;
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int row = (tm << 8) + i * 128 + ha * 64 + wave * 8 + sub;
        if (row > p.M - 1) row = p.M - 1;
        if (p.a_rowmap) row = p.a_rowmap[row];
        a_so[ha][i] = (unsigned)row * (unsigned)p.lda * 2u + (unsigned)(chunk * 16);
      }
  };
  auto stage_advance = [&]() {            // next K-tile; past the last tile of the list the cursor stays (dummy re-loads)
    if (s_kt + 1 < nk) ++s_kt;
    else if (g_stg + nwg < g_end) g_stg += nwg, s_kt = 0, stage_tile(g_stg);
  };
  auto stage = [&](int buf, int slot) {   // two LDS-DMA instructions: rows (i*8 + wave)*8 .. +8 of the half tile
    char* dst = lds + buf * P8_BUF + slot * P8_HALF + wave * 1024;
    const long kb = (long)s_kt * 128;
    if (slot < 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_p8*)(A + kb + a_so[slot][i]), (lds_void_p8*)(dst + i * 8192), 16, 0, 0);
    } else {
      const char* wk = W + kb + b_tile + (long)((slot - 2) * 32) * p.ldw * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_p8*)(wk + (long)(i * 128) * p.ldw * 2 + b_so), (lds_void_p8*)(dst + i * 8192), 16,
                                         0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside a half tile): row (w*? + f*16 + lr), chunk (kk*4 + lq) ^ (lr & 7)
  const int lr = lane & 15, lq = lane >> 4;
  const int a_rd = (wm * 64 + lr) * 128 + ((lq ^ (lr & 7)) << 4);
  const int b_rd = (wn * 32 + lr) * 128 + ((lq ^ (lr & 7)) << 4);
  const int a_rd1 = a_rd ^ 64, b_rd1 = b_rd ^ 64;    // kk = 1: chunk bit 2 flipped

  f32x4 acc[8][4];
  h8 af[4][2], bf[2][2];

  // ---- prologue: 7 half tiles in flight, the first K-tile landed
  stage_tile(g_stg);
  stage(0, P8_SLOT_A0), stage(0, P8_SLOT_B1), stage(0, P8_SLOT_A1), stage(0, P8_SLOT_B0);
  stage_advance();
  stage(1, P8_SLOT_A0), stage(1, P8_SLOT_B1);
  if (SR) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // K-tile 0 has landed; A0 / B1 of K-tile 1 are in flight
  } else {
    stage(1, P8_SLOT_A1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();

  auto read_a = [&](const char* base, int slot) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      af[f][0] = *(const h8*)(base + slot * P8_HALF + f * 2048 + a_rd);
      af[f][1] = *(const h8*)(base + slot * P8_HALF + f * 2048 + a_rd1);
    }
  };
  auto read_b = [&](const char* base, int slot) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      bf[f][0] = *(const h8*)(base + slot * P8_HALF + f * 2048 + b_rd);
      bf[f][1] = *(const h8*)(base + slot * P8_HALF + f * 2048 + b_rd1);
    }
  };
  // one phase: quadrant (HA, HB) of the wave's patch out of K-tile buffer BUF
  auto phase = [&](auto bufc, auto phc) {
    constexpr int BUF = decltype(bufc)::value, PH = decltype(phc)::value;
    constexpr int HA = PH >> 1, HB = (PH == 1 || PH == 2) ? 1 : 0;
    const char* base = lds + BUF * P8_BUF;
    if (PH == 0) read_b(base, P8_SLOT_B0), read_a(base, P8_SLOT_A0);
    if (PH == 1) read_b(base, P8_SLOT_B1);
    if (PH == 2) read_a(base, P8_SLOT_A1);
    if (PH == 3) read_b(base, P8_SLOT_B0);
    if (SR) {      // the stage of this phase, in the read segment (see the SR note above the kernel)
      if (PH == 0) stage(BUF ^ 1, P8_SLOT_A1);
      if (PH == 1) stage(BUF ^ 1, P8_SLOT_B0), stage_advance();
      if (PH == 2) stage(BUF, P8_SLOT_A0);
      if (PH == 3) stage(BUF, P8_SLOT_B1);
    }
    if (PH == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // K-tile +1 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    // plain: the two 32-deep halves of the K-tile; X3: lo.hi, hi.lo, hi.hi of its 32 real k (small terms first)
    constexpr int NTERM = X3 ? 3 : 2;
#pragma unroll
    for (int kk = 0; kk < NTERM; ++kk)
#pragma unroll
      for (int fi = 0; fi < 4; ++fi)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {   // swapped operands -> the accumulator fragment is C^T (see the epilogue)
          const int ka = X3 ? (kk == 0 ? 1 : 0) : kk, kb = X3 ? (kk == 1 ? 1 : 0) : kk;
          acc[HA * 4 + fi][HB * 2 + fj] =
              __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[fj][kb], af[fi][ka], acc[HA * 4 + fi][HB * 2 + fj], 0, 0, 0);
          if (!SR && kk == 0 && fi == 0 && fj == 1) {
            // the stage of this phase, behind the first MFMAs (the matrix pipe is busy while the DMA is issued)
            if (PH == 0) stage(BUF ^ 1, P8_SLOT_B0), stage_advance();
            if (PH == 1) stage(BUF, P8_SLOT_A0);
            if (PH == 2) stage(BUF, P8_SLOT_B1);
            if (PH == 3) stage(BUF, P8_SLOT_A1);
          }
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  };

  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; kt += 2) {
      phase(IC<0>{}, IC<0>{}), phase(IC<0>{}, IC<1>{}), phase(IC<0>{}, IC<2>{}), phase(IC<0>{}, IC<3>{});
      phase(IC<1>{}, IC<0>{}), phase(IC<1>{}, IC<1>{}), phase(IC<1>{}, IC<2>{}), phase(IC<1>{}, IC<3>{});
    }

    asm volatile("" ::: "memory");   // keep the epilogue's loads (bias, residual) BELOW the K-loop: hoisted above it they cost registers
                                     // (spills in the GELU variant) and a vmcnt(0) per tile that drains the DMA prefetch
    // ---- epilogue.  Swapped MFMA operands: lane (lr, lq) of fragment (i, j) owns row i*16 + lr and the 4 CONSECUTIVE
    // columns j*16 + lq*4 .. +3 -> one 16-byte (f32) or 8-byte (f16) store per fragment.  bias, activation, residual at the
    // (row-mapped) destination row, as gemm_kernel does.
    int tm, tn;
    decode(g_cmp, tm, tn);
    const int m0 = tm << 8, n0 = tn << 8;
    const int colbase = n0 + wn * 64 + lq * 4;
    float4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bv[j] = p.bias ? *(const float4*)(p.bias + colbase + (j >> 1) * 32 + (j & 1) * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    // The residual of RB fragment rows (RB x 4 fragments x 16 B per lane) is loaded in ONE batch into the registers the
    // operand fragments no longer need: the epilogue waits for 8 / RB memory round trips per tile instead of 32.
    constexpr int RB = P8_EPI_ROWS;
#pragma unroll
    for (int hb = 0; hb < 8 / RB; ++hb) {
      int drow[RB];
      float4 rv[RB][4];
#pragma unroll
      for (int ii = 0; ii < RB; ++ii) {
        const int row = m0 + wm * 128 + ((hb * RB + ii) >> 2) * 64 + ((hb * RB + ii) & 3) * 16 + lr;
        int d = row < p.M ? row : -1;
        if (p.rowmap && d >= 0) d = p.rowmap[row];
        drow[ii] = d;
      }
      if (p.res) {
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) {
          const int dr = drow[ii] < 0 ? 0 : drow[ii];
          const float* rp = p.res + (long)(p.res_mod > 0 ? dr % p.res_mod : dr) * p.ldr + colbase;
#pragma unroll
          for (int j = 0; j < 4; ++j) rv[ii][j] = *(const float4*)(rp + (j >> 1) * 32 + (j & 1) * 16);
        }
      }
#pragma unroll
      for (int ii = 0; ii < RB; ++ii) {
        const int i = hb * RB + ii;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cj = (j >> 1) * 32 + (j & 1) * 16;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
          v[0] += bv[j].x, v[1] += bv[j].y, v[2] += bv[j].z, v[3] += bv[j].w;
          if (ACT == ACT_GELU) {
            if (OUT == 1) {        // fp16 result: the transcendental-free polynomial, two elements per packed instruction
              const f32x2_g g0 = gelu_poly2((f32x2_g){v[0], v[1]}), g1 = gelu_poly2((f32x2_g){v[2], v[3]});
              v[0] = g0[0], v[1] = g0[1], v[2] = g1[0], v[3] = g1[1];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = gelu_fast(v[r]);
            }
          }
          if (p.res) v[0] += rv[ii][j].x, v[1] += rv[ii][j].y, v[2] += rv[ii][j].z, v[3] += rv[ii][j].w;
          if (drow[ii] >= 0) {
            if (OUT == 2) {        // x3 row: the 4 columns lie inside one 32-block (colbase + cj is a multiple of 4)
              h4 hi, lo;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                half_t a, b;
                split_f16(v[r], a, b);
                hi[r] = a, lo[r] = b;
              }
              half_t* cp = (half_t*)p.C + (long)drow[ii] * p.ldc + x3_col(colbase + cj);
              *(h4*)cp = hi;
              *(h4*)(cp + 32) = lo;
            } else if (OUT == 1) {
              *(h4*)((half_t*)p.C + (long)drow[ii] * p.ldc + colbase + cj) =
                  (h4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            } else {
              f32x4 o = (f32x4){v[0], v[1], v[2], v[3]};
              __builtin_nontemporal_store(o, (f32x4*)((float*)p.C + (long)drow[ii] * p.ldc + colbase + cj));
            }
          }
        }
      }
    }
    g_cmp += nwg;
    if (g_cmp >= g_end) break;
  }
  if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy tail stages must land before the LDS is handed on
}

// returns SAMPT_ERR_UNSUPPORTED when the shape does not fit this kernel
int gemm_f16_p8_launch(const GemmP& p, hipStream_t s) {
  if (p.conv || p.w_kn || p.nb1 * p.nb2 != 1 || (p.K % 128) || p.M < 256 || (p.N % 256)) return SAMPT_ERR_UNSUPPORTED;
  if ((p.lda % 8) || (p.ldw % 8) || (((uintptr_t)p.A | (uintptr_t)p.W) & 15)) return SAMPT_ERR_UNSUPPORTED;
  if ((p.ldc % 4) || (p.res && (p.ldr % 4))) return SAMPT_ERR_UNSUPPORTED;
  if (p.out_f16 == 2 && (p.ldc % 64)) return SAMPT_ERR_UNSUPPORTED;
  // 32-bit byte offsets inside the operands (with a_rowmap the caller guarantees the gathered matrix is < 4 GiB as well)
  if ((double)p.M * p.lda * 2.0 >= 4294967296.0 || (double)p.N * p.ldw * 2.0 >= 4294967296.0) return SAMPT_ERR_UNSUPPORTED;
  if (p.act != ACT_NONE && p.act != ACT_GELU) return SAMPT_ERR_UNSUPPORTED;
  GemmP q = p;
  const int nt_m = cdiv(p.M, 256), nt_n = p.N / 256;
  int R = 4;                      // strip height in row panels (2 / 8 measured: more L2 misses, profiles/r3_gemm_hbm_traffic_strip*)
  if (R > nt_m) R = nt_m;
  q.xcd_swizzle = R;
  if (q.p8_stagger == 0) q.p8_stagger = g_p8_stagger;
  const long ntiles = (long)nt_m * nt_n;
  // workgroups per XCD: one per CU (32) by default; fewer leaves whole CUs to kernels of other streams (a 512-thread,
  // 128-KiB workgroup owns its CU: nothing else becomes resident beside it)
  const int wgs = p.p8_wgs >= 1 && p.p8_wgs <= 32 ? p.p8_wgs : 32;
  int per_xcd = (int)((ntiles + 7) / 8);
  if (per_xcd > wgs) {
    // An XCD's tiles take ceil(tiles / workgroups) rounds whatever the remainder: the smallest workgroup count with the SAME number
    // of rounds finishes at the same time and leaves the other CUs to whoever runs beside the launch (ViT-H, 8 frames: proj / fc2
    // have 80 tiles per XCD = 3 rounds on 30 workgroups and on 27; the tracker's window chain then ends 15 ms earlier beside an
    // unchanged encoder, profiles/r6_c22_*).  Same tiles, same results.
    const int tpx = per_xcd, rounds = cdiv(tpx, wgs);
    per_xcd = wgs;
    if (g_p8_trim)
      while (per_xcd > 1 && cdiv(tpx, per_xcd - 1) == rounds) --per_xcd;
  }
  const dim3 grid(8 * per_xcd), block(512);
#define P8_LAUNCH(AC, OU, X)                                                                \
  do {                                                                                      \
    if (g_p8_sched == 0) hipLaunchKernelGGL((gemm_f16_p8<AC, OU, X, true>), grid, block, 0, s, q); \
    else hipLaunchKernelGGL((gemm_f16_p8<AC, OU, X, false>), grid, block, 0, s, q);         \
  } while (0)
  const int out = p.out_f16;      // 0 f32, 1 f16, 2 x3 rows
  if (p.x3) {
    if (p.act == ACT_GELU) { if (out == 2) P8_LAUNCH(ACT_GELU, 2, true); else return SAMPT_ERR_UNSUPPORTED; }
    else if (out == 2) P8_LAUNCH(ACT_NONE, 2, true);
    else if (out == 0) P8_LAUNCH(ACT_NONE, 0, true);
    else return SAMPT_ERR_UNSUPPORTED;
  } else {
    if (out == 2) return SAMPT_ERR_UNSUPPORTED;
    if (p.act == ACT_GELU) { if (out == 1) P8_LAUNCH(ACT_GELU, 1, false); else P8_LAUNCH(ACT_GELU, 0, false); }
    else if (out == 1) P8_LAUNCH(ACT_NONE, 1, false);
    else P8_LAUNCH(ACT_NONE, 0, false);
  }
#undef P8_LAUNCH
  SAMPT_CHECK_LAUNCH("gemm_f16_p8");
  return SAMPT_OK;
}

}  // namespace sampt
