// fp16 MFMA GEMM for the ViT image encoder, LDS-DMA version (the dominant kernel of the SAM-PT hot path).
//
//   C[M][N] = epi(A[M][K] . W[N][K]^T + bias) (+ residual)        A, W fp16 K-contiguous; fp32 accumulate
//
// 128 x 128 x 64 block tile, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 fragments of
// v_mfma_f32_16x16x32_f16.  Both operand slabs go HBM -> LDS with `global_load_lds_dwordx4` (no VGPR staging, no
// ds_write pass): each wave-instruction lands 8 rows x 128 B.  The LDS image is linear per instruction (hardware
// rule: M0 base + lane*16), so the bank-conflict swizzle is applied on the SOURCE side: the 16-byte chunk that lane l
// fetches for tile row r is chunk (l&7) ^ (r&7), and fragment reads apply the same XOR.  Two LDS buffers (64 KiB
// total, 2 workgroups per CU) and ONE barrier per K-slab: the DMA of slab k+1 is issued right after the barrier
// that retires slab k and overlaps its 32 MFMAs per wave.
//
// Edge handling: rows beyond M / N are clamped to the last valid row (their products land in accumulator rows /
// columns the epilogue never stores); K must be a multiple of 64 (true for every ViT GEMM: 768..5120).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace sampt {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// MF = 16: v_mfma_f32_16x16x32_f16 fragments; MF = 32: v_mfma_f32_32x32x16_f16 (each operand element feeds 32 instead of
// 16 products: half the operand-register reads per FLOP, which matters on a power-limited part).  The source-side
// swizzle differs with the fragment read pattern: chunk ^= row & 7 (MF 16) or chunk ^= (row >> 1) & 7 (MF 32), both
// conflict-free for their ds_read_b128 lane groups.
template <int BM, int BN, int NBUF, int WTM, int WTN, int MF, int OCC = (NBUF == 1 ? 4 : 2)>
__global__ __launch_bounds__((BM / WTM) * (BN / WTN) * 64, OCC) void gemm_f16_glds(GemmP p) {
  constexpr int BK = 64;
  constexpr int NWM = BM / WTM, NWN = BN / WTN, NWAVES = NWM * NWN;   // waves: NWM x NWN, each a WTM x WTN sub-tile
  constexpr int A_IT = BM / (8 * NWAVES), B_IT = BN / (8 * NWAVES);   // 8-row DMA pieces per wave
  constexpr int FM = WTM / MF, FN = WTN / MF;
  typedef typename std::conditional<MF == 16, f32x4, f32x16>::type acc_t;
  __shared__ __attribute__((aligned(1024))) half_t lds[NBUF * (BM + BN) * BK];  // ONE object: [buf][A rows | B rows][64]
  half_t* As0 = lds;
  half_t* Bs0 = lds + BM * BK;
  constexpr int BUF = (BM + BN) * BK;

  // `wave` is wave-uniform but derived from threadIdx: tell the compiler (readfirstlane), otherwise every LDS destination
  // of the DMA (M0 values) and every per-wave base lives in a VECTOR register and is re-broadcast before each use
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  // Persistent mode (p.persist): the grid is one generation of workgroups (4 per CU) and each walks its XCD's tile
  // sequence with a stride of one generation, so the ~128 tiles co-resident on an XCD start together and stay within a
  // few K-slabs of each other: their working set (8 A panels + 16 W tiles, one slab deep) then fits the 4 MiB L2, which
  // the free-running order does not (measured: 7.8x -> see profiles/r1_gemm_hbm_traffic.json).
  const int idx_step = p.persist ? (int)(gridDim.x >> 3) : 0x40000000;
  for (int idx = blockIdx.x >> 3, first = 1; idx < (p.persist ? p.persist : 0x40000000); idx += idx_step, first = 0) {
  int tile_m, tile_n;
  if (p.xcd_swizzle) {
    // XCD-aware tile order (blocks are dispatched round-robin over the 8 XCDs, each with a private 4 MiB L2): XCD x
    // owns the strips {x, x+8, ...} of 8 consecutive row panels and walks a strip column-major, so the ~128 tiles
    // resident on one XCD form an 8 x 16 patch that shares A panels and W tiles through that XCD's L2.  Pure speed:
    // the mapping is a bijection over the tiles whatever the real placement is.
    const int nt_m = (p.M + BM - 1) / BM, nt_n = (p.N + BN - 1) / BN;
    const int R = p.xcd_swizzle;                    // strip height in row panels (8, or less for small M)
    const int nstrips = (nt_m + R - 1) / R, per_strip = R * nt_n;
    const int xcd = blockIdx.x & 7;
    const int j = idx / per_strip, t = idx - j * per_strip;
    const int strip = xcd + 8 * j;
    if (strip >= nstrips) break;
    const int rows = min(R, nt_m - R * strip);
    tile_n = t / rows;
    if (tile_n >= nt_n) {
      if (p.persist) continue;
      break;
    }
    tile_m = strip * R + (t - tile_n * rows);
  } else {
    tile_m = blockIdx.y, tile_n = blockIdx.x;
  }
  if (!first) __syncthreads();  // every wave is done with the previous tile's LDS slab
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const half_t* __restrict__ A = (const half_t*)p.A;
  const half_t* __restrict__ W = (const half_t*)p.W;

  // per-lane DMA source offsets (row clamped, chunk XOR-swizzled); piece i of this wave = tile rows (wave*IT+i)*8..+8.
  // One 32-bit BYTE offset per piece against the uniform operand base (operands are < 4 GiB): half the registers of a
  // pointer per piece, and the address is formed by the load's own scalar-base + vector-offset mode.
  const int sub = lane >> 3;
  unsigned a_boff[A_IT], b_boff[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int trow = (wave * A_IT + i) * 8 + sub;               // row inside the tile
    const int chunk = (lane & 7) ^ (MF == 16 ? (trow & 7) : ((trow >> 1) & 7));
    int row = m0 + trow;
    if (row > p.M - 1) row = p.M - 1;
    if (p.a_rowmap) row = p.a_rowmap[row];
    a_boff[i] = ((unsigned)row * (unsigned)p.lda + (unsigned)(chunk * 8)) * 2u;
  }
  // BUNI (the 160-wide tile; its launcher guarantees N % BN == 0, so no row is ever clamped): the pieces of a wave are 8
  // weight rows apart and (trow & 7) == sub for every piece, so ONE vector offset serves all of them and the piece stride
  // goes into the scalar base — B_IT - 1 registers less in a kernel that has none to spare
  constexpr bool BUNI = BN == 160;
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int trow = (wave * B_IT + i) * 8 + sub;
    const int chunk = (lane & 7) ^ (MF == 16 ? (trow & 7) : ((trow >> 1) & 7));
    int row = n0 + trow;
    if (row > p.N - 1) row = p.N - 1;
    b_boff[i] = ((unsigned)row * (unsigned)p.ldw + (unsigned)(chunk * 8)) * 2u;
  }
  const size_t b_piece = (size_t)8 * p.ldw * 2;                 // bytes between consecutive pieces (uniform)
  auto issue = [&](int kt, int buf) {
    half_t* ab = As0 + buf * BUF + wave * A_IT * 8 * BK;
    half_t* bb = Bs0 + buf * BUF + wave * B_IT * 8 * BK;
    const char* Ak = (const char*)A + (size_t)kt * (BK * 2);    // uniform
    const char* Wk = (const char*)W + (size_t)kt * (BK * 2);
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(Ak + a_boff[i]), (lds_void*)(ab + i * 8 * BK), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      if constexpr (BUNI)
        __builtin_amdgcn_global_load_lds((glb_void*)(Wk + i * b_piece + b_boff[0]), (lds_void*)(bb + i * 8 * BK), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((glb_void*)(Wk + b_boff[i]), (lds_void*)(bb + i * 8 * BK), 16, 0, 0);
    }
  };

  acc_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < (MF == 16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  // fragment reads: lane (lr, lq) takes row lr of the fragment and the 16-byte K-chunk kk*KQ + lq of that row
  // (MF 16: lr = lane & 15, 4 chunks per 32-wide K step; MF 32: lr = lane & 31, 2 chunks per 16-wide K step);
  // offsets in halfs inside a buffer: row*64 + ((kk*KQ + lq) ^ swz(row))*8 ; swz(row) is the same for every fragment
  constexpr int KQ = MF == 16 ? 4 : 2;
  const int lr = lane & (MF - 1), lq = lane / MF;
  int a_off[FM], b_off[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) a_off[i] = (wm * WTM + i * MF + lr) * BK;
#pragma unroll
  for (int j = 0; j < FN; ++j) b_off[j] = BM * BK + (wn * WTN + j * MF + lr) * BK;
  const int sw = MF == 16 ? (lr & 7) : ((lr >> 1) & 7);

  if (NBUF == 2) issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = NBUF == 2 ? (kt & 1) : 0;
    if (NBUF == 1) {
      // single buffer, 4 workgroups per CU: the other resident workgroups compute while this one waits for its DMA
      if (kt > 0) __syncthreads();      // everyone done reading the previous slab
      issue(kt, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (NBUF == 2 && kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const half_t* base = lds + buf * BUF;
#pragma unroll
    for (int kk = 0; kk < 8 / KQ; ++kk) {
      const int pos = ((kk * KQ + lq) ^ sw) * 8;
      if constexpr (FN > 4) {
        // wide wave tiles (FN = 5): keep the A fragments of the K-step and ONE B fragment (plus the next one in flight)
        // live instead of all FN — the 80 accumulator registers leave no room for 9 operand fragments under the
        // 128-register budget of 4 workgroups per CU (the scheduler barrier keeps the compiler from hoisting the loads)
        h8 a[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = *(const h8*)(base + a_off[i] + pos);
        h8 bc = *(const h8*)(base + b_off[0] + pos);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          h8 bn = bc;
          if (j + 1 < FN) bn = *(const h8*)(base + b_off[j + 1] + pos);
#pragma unroll
          for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bc, a[i], acc[i][j], 0, 0, 0);
          bc = bn;
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      h8 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *(const h8*)(base + a_off[i] + pos);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *(const h8*)(base + b_off[j] + pos);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {  // swapped operands -> D^T: see epilogue
          if constexpr (MF == 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue.  The MFMAs were issued with swapped operands (W fragment as A, A fragment as B), so each accumulator
  // fragment holds C^T: lane (lr, lq) owns row m = lr and the 4 CONSECUTIVE columns lq*4..+3 -> one 16-byte (f32) or
  // 8-byte (f16) store per fragment instead of four scattered scalar stores.  Same contract as gemm_kernel: bias,
  // activation, residual at the (row-mapped) destination row.
  // (MF 32: the 16 accumulator registers are 4 such groups, columns 8*g + 4*lq .. +3 of the fragment.)
  // The launcher guarantees N, ldc, ldr multiples of 4 (16-byte rows), so every store below is one vector.
  constexpr int NG = MF == 16 ? 1 : 4;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int row = m0 + wm * WTM + i * MF + lr;
    if (row >= p.M) continue;
    int drow = p.rowmap ? p.rowmap[row] : row;
    if (drow < 0) continue;
    int rrow = p.res_mod > 0 ? drow % p.res_mod : drow;
#pragma unroll
    for (int jg = 0; jg < FN * NG; ++jg) {
      const int j = jg / NG, g = jg % NG;
      int col = n0 + wn * WTN + j * MF + (MF == 16 ? lq * 4 : 8 * g + 4 * lq);
      if (col >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r] * p.alpha;
      if (p.bias) {
        float4 bv = *(const float4*)(p.bias + col);
        v[0] += bv.x, v[1] += bv.y, v[2] += bv.z, v[3] += bv.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
      if (p.res) {
        float4 rv = *(const float4*)(p.res + (long)rrow * p.ldr + col);
        v[0] += rv.x, v[1] += rv.y, v[2] += rv.z, v[3] += rv.w;
      }
      if (p.out_f16) {
        h4 o = (h4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *(h4*)((half_t*)p.C + (long)drow * p.ldc + col) = o;
      } else {
        // fp32 C (64-byte row segments per store group) is streamed out non-temporally: +3..8 % on the proj / fc2 shapes;
        // the 32-byte segments of fp16 C need the L2's write combining and lose with it (measured both ways)
        f32x4 o = (f32x4){v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(o, (f32x4*)((float*)p.C + (long)drow * p.ldc + col));
      }
    }
  }
  if (!p.xcd_swizzle) break;
  }  // tile loop
}

// ---------------------------------------------------------------------------------------------
// Experimental (SAMPT_GEMM_VARIANT=9): BK = 32 slabs, double-buffered, still 32 KiB of LDS -> 4 workgroups per CU AND a
// slab of prefetch per workgroup.  128 x 128 tile, 4 waves x (64 x 64), one v_mfma_f32_16x16x32_f16 K-step per slab.
// LDS rows are 64 B (4 chunks of 16 B); a DMA piece covers 16 rows; chunk swizzle pos = chunk ^ F[(row >> 2) & 3],
// F = {0, 3, 2, 1}: every ds_read_b128 lane group then touches 16 distinct 16-byte slots of the 256-B bank row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void gemm_f16_glds_bk32(GemmP p) {
  constexpr int BM = 128, BN = 128, BK = 32, FM = 4, FN = 4, BUF = (BM + BN) * BK;
  __shared__ __attribute__((aligned(1024))) half_t lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nt_m = (p.M + BM - 1) / BM, nt_n = (p.N + BN - 1) / BN;
  const int R = p.xcd_swizzle, nstrips = (nt_m + R - 1) / R, per_strip = R * nt_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int j = idx / per_strip, t = idx - j * per_strip, strip = xcd + 8 * j;
  if (strip >= nstrips) return;
  const int rows = min(R, nt_m - R * strip);
  const int tile_n = t / rows;
  if (tile_n >= nt_n) return;
  const int tile_m = strip * R + (t - tile_n * rows);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const half_t* __restrict__ A = (const half_t*)p.A;
  const half_t* __restrict__ W = (const half_t*)p.W;
  const int prow = lane >> 2, ppos = lane & 3;
  const half_t* a_src[2];
  const half_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int trow = (wave * 2 + i) * 16 + prow;
    const int f = (trow >> 2) & 3, chunk = ppos ^ ((4 - f) & 3);          // F = {0, 3, 2, 1}
    int row = min(m0 + trow, p.M - 1);
    if (p.a_rowmap) row = p.a_rowmap[row];
    a_src[i] = A + (long)row * p.lda + chunk * 8;
    b_src[i] = W + (long)min(n0 + trow, p.N - 1) * p.ldw + chunk * 8;
  }
  auto issue = [&](int kt, int buf) {
    half_t* ab = lds + buf * BUF + wave * 2 * 16 * BK;
    half_t* bb = lds + buf * BUF + BM * BK + wave * 2 * 16 * BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void*)(a_src[i] + kt * BK), (lds_void*)(ab + i * 16 * BK), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(b_src[i] + kt * BK), (lds_void*)(bb + i * 16 * BK), 16, 0, 0);
    }
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / BK;
  const int lr = lane & 15, lq = lane >> 4;
  const int fsw = (lr >> 2) & 3, pos = (lq ^ ((4 - fsw) & 3)) * 8;
  int a_off[FM], b_off[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) a_off[i] = (wm * 64 + i * 16 + lr) * BK + pos;
#pragma unroll
  for (int jj = 0; jj < FN; ++jj) b_off[jj] = BM * BK + (wn * 64 + jj * 16 + lr) * BK + pos;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                      // slab kt landed everywhere; everyone is done reading slab kt-1
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const half_t* base = lds + buf * BUF;
    h8 a[FM], b[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) a[i] = *(const h8*)(base + a_off[i]);
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) b[jj] = *(const h8*)(base + b_off[jj]);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int jj = 0; jj < FN; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[jj], a[i], acc[i][jj], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int row = m0 + wm * 64 + i * 16 + lr;
    if (row >= p.M) continue;
    int drow = p.rowmap ? p.rowmap[row] : row;
    if (drow < 0) continue;
    int rrow = p.res_mod > 0 ? drow % p.res_mod : drow;
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) {
      int col = n0 + wn * 64 + jj * 16 + lq * 4;
      if (col >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][jj][r] * p.alpha;
      if (p.bias) {
        float4 bv = *(const float4*)(p.bias + col);
        v[0] += bv.x, v[1] += bv.y, v[2] += bv.z, v[3] += bv.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
      if (p.res) {
        float4 rv = *(const float4*)(p.res + (long)rrow * p.ldr + col);
        v[0] += rv.x, v[1] += rv.y, v[2] += rv.z, v[3] += rv.w;
      }
      if (p.out_f16) {
        *(h4*)((half_t*)p.C + (long)drow * p.ldc + col) = (h4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      } else {
        f32x4 o = (f32x4){v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(o, (f32x4*)((float*)p.C + (long)drow * p.ldc + col));
      }
    }
  }
}

int gemm_f16_p8_launch(const GemmP& p, hipStream_t s);   // gemm_f16_p8.hip

// returns SAMPT_ERR_UNSUPPORTED when the shape does not fit this kernel (caller falls back to gemm_kernel)
int gemm_f16_glds_launch(const GemmP& p, hipStream_t s) {
  static const int p8 = getenv("SAMPT_GEMM_P8") ? atoi(getenv("SAMPT_GEMM_P8")) : 1;
  if (p8) {
    int rc = gemm_f16_p8_launch(p, s);
    if (rc != SAMPT_ERR_UNSUPPORTED) return rc;
  }
  if (p.conv || p.w_kn || p.nb1 * p.nb2 != 1 || p.K % 64 || p.M < 128 || p.N < 128) return SAMPT_ERR_UNSUPPORTED;
  if ((p.lda % 8) || (p.ldw % 8) || (((uintptr_t)p.A | (uintptr_t)p.W) & 15)) return SAMPT_ERR_UNSUPPORTED;
  if ((p.N % 4) || (p.ldc % 4) || (p.res && (p.ldr % 4))) return SAMPT_ERR_UNSUPPORTED;   // vector epilogue only
  GemmP q = p;
  static const int swz = getenv("SAMPT_GEMM_SWZ") ? atoi(getenv("SAMPT_GEMM_SWZ")) : 1;
  q.xcd_swizzle = swz;
  dim3 grid(cdiv(p.N, 128), cdiv(p.M, 128), 1), block(256);
  if (swz) {
    const int nt_m = cdiv(p.M, 128), nt_n = cdiv(p.N, 128);
    int R = 8;
    while (R > 1 && cdiv(nt_m, R) < 16) R /= 2;     // keep >= 2 strips per XCD so that all 8 XCDs get work
    const int nstrips = cdiv(nt_m, R), per_xcd = cdiv(nstrips, 8);
    q.xcd_swizzle = R;
    grid = dim3(8 * per_xcd * R * nt_n, 1, 1);
  }
  static const int variant = getenv("SAMPT_GEMM_VARIANT") ? atoi(getenv("SAMPT_GEMM_VARIANT")) : 1;
  // 128 x 160 tiles wherever N is a multiple of 160 (every ViT-H GEMM): measured +3.6 .. +4.8 % per shape and +1 % end to
  // end over the 128 x 128 tile (profiles/r2_v4_gemm_microbench_{default,bn160}.log); SAMPT_GEMM_BN160=0 disables it
  static const bool bn160 = !(getenv("SAMPT_GEMM_BN160") && atoi(getenv("SAMPT_GEMM_BN160")) == 0);
  if ((variant == 4 || variant == 5 || variant == 7) && swz && p.M >= 256 && p.N >= 256) {
    // 256 x 256 tile, 8 waves of 128 x 64, two 64 KiB LDS stages (1 workgroup per CU): a K-slab is 64 MFMAs per wave, so
    // the DMA of the next slab has ~2000 cycles to land and LDS traffic per FLOP halves against the 128 x 128 tile
    const int nt_m = cdiv(p.M, 256), nt_n = cdiv(p.N, 256);
    int R = 4;
    while (R > 1 && cdiv(nt_m, R) < 16) R /= 2;
    q.xcd_swizzle = R;
    grid = dim3(8 * cdiv(cdiv(nt_m, R), 8) * R * nt_n, 1, 1);
    if (variant == 7) hipLaunchKernelGGL((gemm_f16_glds<256, 256, 2, 128, 128, 16, 1>), grid, dim3(256), 0, s, q);
    else if (variant == 4) hipLaunchKernelGGL((gemm_f16_glds<256, 256, 2, 128, 64, 16>), grid, dim3(512), 0, s, q);
    else hipLaunchKernelGGL((gemm_f16_glds<256, 256, 2, 128, 64, 32>), grid, dim3(512), 0, s, q);
  } else if ((variant == 10 || (variant == 1 && bn160)) && swz && p.N % 160 == 0) {
    // 128 x 160 tile (4 waves x (64 x 80)): the ViT-H / ViT-L widths are multiples of 160, and with 160-wide column tiles
    // every encoder GEMM is a whole number of generations of 1024 resident workgroups (N = 1280: 2048 tiles = 2.0
    // generations instead of 2.5; 3840: 6.0 instead of 7.5; 5120: 8.0 instead of 10.0) — no half-empty last wave; LDS is
    // 36 KiB per workgroup, still 4 per CU; a wave's slab costs 18 KiB of fragment reads per 40 MFMAs instead of 16 per 32.
    const int nt_m = cdiv(p.M, 128), nt_n = p.N / 160;
    int R = 8;
    while (R > 1 && cdiv(nt_m, R) < 16) R /= 2;
    q.xcd_swizzle = R;
    grid = dim3(8 * cdiv(cdiv(nt_m, R), 8) * R * nt_n, 1, 1);
    hipLaunchKernelGGL((gemm_f16_glds<128, 160, 1, 64, 80, 16>), grid, block, 0, s, q);
  } else if (variant == 6) {
    hipLaunchKernelGGL((gemm_f16_glds<128, 128, 1, 64, 64, 32>), grid, block, 0, s, q);
  } else if (variant == 3 && swz && p.M >= 256) {
    const int nt_m = cdiv(p.M, 256), nt_n = cdiv(p.N, 128);
    int R = 4;
    while (R > 1 && cdiv(nt_m, R) < 16) R /= 2;
    q.xcd_swizzle = R;
    grid = dim3(8 * cdiv(cdiv(nt_m, R), 8) * R * nt_n, 1, 1);
    hipLaunchKernelGGL((gemm_f16_glds<256, 128, 1, 64, 64, 16>), grid, dim3(512), 0, s, q);
  } else if (variant == 9 && swz) {
    hipLaunchKernelGGL(gemm_f16_glds_bk32, grid, block, 0, s, q);
  } else if (variant == 8 && swz) {
    // persistent: one generation = 4 workgroups per CU x 256 CUs = 128 per XCD
    const int nt_m = cdiv(p.M, 128), nt_n = cdiv(p.N, 128);
    const int R = q.xcd_swizzle, per_xcd = cdiv(cdiv(nt_m, R), 8) * R * nt_n;
    q.persist = per_xcd;
    const int gen = per_xcd < 128 ? per_xcd : 128;
    hipLaunchKernelGGL((gemm_f16_glds<128, 128, 1, 64, 64, 16>), dim3(8 * gen), block, 0, s, q);
  } else if (variant == 1 || variant == 3) {
    // experiment knob: SAMPT_GEMM_LDS_PAD=<bytes> of unused dynamic LDS per workgroup, e.g. 9728 makes a
    // workgroup cost 41.5 KiB so that only 3 (not 4) fit a CU and ~35 KiB + wave slots stay free for the tracker's kernels
    static const int lds_pad = getenv("SAMPT_GEMM_LDS_PAD") ? atoi(getenv("SAMPT_GEMM_LDS_PAD")) : 0;
    hipLaunchKernelGGL((gemm_f16_glds<128, 128, 1, 64, 64, 16>), grid, block,
                       (size_t)(lds_pad > 0 && lds_pad <= 32768 ? lds_pad : 0), s, q);
  }
  else hipLaunchKernelGGL((gemm_f16_glds<128, 128, 2, 64, 64, 16>), grid, block, 0, s, q);
  SAMPT_CHECK_LAUNCH("gemm_f16_glds");
  return SAMPT_OK;
}

}  // namespace sampt
