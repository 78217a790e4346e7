// fp16 MFMA GEMM, LDS-DMA version with 128-row tiles: the FALLBACK of the ViT encoder's GEMM for the shapes the 256 x 256
// 8-phase persistent kernel (gemm_f16_p8.hip, the dominant kernel of the SAM-PT hot path) does not take — fewer than 256
// rows, N not a multiple of 256, K not a multiple of 128 (reduced test geometries, odd batches) — and the round-1 / round-2
// kernel every ViT GEMM ran on (A/B: profiles/r3_v1_gemm_microbench_old.log).
//
//   C[M][N] = epi(A[M][K] . W[N][K]^T + bias) (+ residual)        A, W fp16 K-contiguous; fp32 accumulate
//
// 128 x 128 x 64 (or 128 x 160 x 64 when N % 160 == 0) block tile, 256 threads = 4 waves (2 x 2), each wave 64 x 64 (64 x 80)
// = 4 x 4 (4 x 5) fragments of v_mfma_f32_16x16x32_f16.  Both operand slabs go HBM -> LDS with `global_load_lds_dwordx4` (no
// VGPR staging, no ds_write pass): each wave-instruction lands 8 rows x 128 B.  The LDS image is linear per instruction
// (hardware rule: M0 base + lane*16), so the bank-conflict swizzle is applied on the SOURCE side: the 16-byte chunk that
// lane l fetches for tile row r is chunk (l&7) ^ (r&7), and fragment reads apply the same XOR.  ONE 32 / 36 KiB LDS buffer
// and four workgroups per CU: the other resident workgroups multiply while this one waits for its DMA (two barriers per
// K-slab: this structure tops out near 0.3 of the fp16 peak, which is why the 8-phase kernel replaced it).
//
// Edge handling: rows beyond M / N are clamped to the last valid row (their products land in accumulator rows /
// columns the epilogue never stores); K must be a multiple of 64 (true for every ViT GEMM: 768..5120).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace sampt {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int BM, int BN, int WTM, int WTN>
__global__ __launch_bounds__((BM / WTM) * (BN / WTN) * 64, 4) void gemm_f16_glds(GemmP p) {
  constexpr int BK = 64, MF = 16;
  constexpr int NWM = BM / WTM, NWN = BN / WTN, NWAVES = NWM * NWN;   // waves: NWM x NWN, each a WTM x WTN sub-tile
  constexpr int A_IT = BM / (8 * NWAVES), B_IT = BN / (8 * NWAVES);   // 8-row DMA pieces per wave
  constexpr int FM = WTM / MF, FN = WTN / MF;
  __shared__ __attribute__((aligned(1024))) half_t lds[(BM + BN) * BK];  // ONE object: [A rows | B rows][64]
  half_t* As0 = lds;
  half_t* Bs0 = lds + BM * BK;

  // `wave` is wave-uniform but derived from threadIdx: tell the compiler (readfirstlane), otherwise every LDS destination
  // of the DMA (M0 values) and every per-wave base lives in a VECTOR register and is re-broadcast before each use
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  // XCD-aware tile order (blocks are dispatched round-robin over the 8 XCDs, each with a private 4 MiB L2): XCD x owns the
  // strips {x, x+8, ...} of R consecutive row panels and walks a strip column-major, so the tiles resident on one XCD form a
  // patch that shares A panels and W tiles through that XCD's L2.  Pure speed: the mapping is a bijection over the tiles
  // whatever the real placement is.
  const int nt_m = (p.M + BM - 1) / BM, nt_n = (p.N + BN - 1) / BN;
  const int R = p.xcd_swizzle;                    // strip height in row panels (8, or less for small M)
  const int nstrips = (nt_m + R - 1) / R, per_strip = R * nt_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int sj = idx / per_strip, t = idx - sj * per_strip;
  const int strip = xcd + 8 * sj;
  if (strip >= nstrips) return;
  const int rows = min(R, nt_m - R * strip);
  const int tile_n = t / rows;
  if (tile_n >= nt_n) return;
  const int tile_m = strip * R + (t - tile_n * rows);
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
    const int chunk = (lane & 7) ^ (trow & 7);
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
    const int chunk = (lane & 7) ^ (trow & 7);
    int row = n0 + trow;
    if (row > p.N - 1) row = p.N - 1;
    b_boff[i] = ((unsigned)row * (unsigned)p.ldw + (unsigned)(chunk * 8)) * 2u;
  }
  const size_t b_piece = (size_t)8 * p.ldw * 2;                 // bytes between consecutive pieces (uniform)
  auto issue = [&](int kt) {
    half_t* ab = As0 + wave * A_IT * 8 * BK;
    half_t* bb = Bs0 + wave * B_IT * 8 * BK;
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

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  // fragment reads: lane (lr, lq) takes row lr of the fragment and the 16-byte K-chunk kk*4 + lq of that row;
  // offsets in halfs: row*64 + ((kk*4 + lq) ^ (row & 7))*8 ; the swizzle term is the same for every fragment
  const int lr = lane & 15, lq = lane >> 4;
  int a_off[FM], b_off[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) a_off[i] = (wm * WTM + i * MF + lr) * BK;
#pragma unroll
  for (int j = 0; j < FN; ++j) b_off[j] = BM * BK + (wn * WTN + j * MF + lr) * BK;
  const int sw = lr & 7;

  for (int kt = 0; kt < nk; ++kt) {
    // single buffer, 4 workgroups per CU: the other resident workgroups compute while this one waits for its DMA
    if (kt > 0) __syncthreads();      // everyone done reading the previous slab
    issue(kt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const half_t* base = lds;
    // plain: the two 32-deep halves of the slab; x3 (GemmP::x3): lo.hi, hi.lo, hi.hi of the slab's 32 real k
    const int nterm = p.x3 ? 3 : 2;
    for (int kk = 0; kk < nterm; ++kk) {
      const int ka = p.x3 ? (kk == 0 ? 1 : 0) : kk, kb = p.x3 ? (kk == 1 ? 1 : 0) : kk;
      const int pos = ((ka * 4 + lq) ^ sw) * 8, posb = ((kb * 4 + lq) ^ sw) * 8;
      if constexpr (FN > 4) {
        // wide wave tiles (FN = 5): keep the A fragments of the K-step and ONE B fragment (plus the next one in flight)
        // live instead of all FN — the 80 accumulator registers leave no room for 9 operand fragments under the
        // 128-register budget of 4 workgroups per CU (the scheduler barrier keeps the compiler from hoisting the loads)
        h8 a[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = *(const h8*)(base + a_off[i] + pos);
        h8 bc = *(const h8*)(base + b_off[0] + posb);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          h8 bn = bc;
          if (j + 1 < FN) bn = *(const h8*)(base + b_off[j + 1] + posb);
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
        for (int j = 0; j < FN; ++j) b[j] = *(const h8*)(base + b_off[j] + posb);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)   // swapped operands -> D^T: see epilogue
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue.  The MFMAs were issued with swapped operands (W fragment as A, A fragment as B), so each accumulator
  // fragment holds C^T: lane (lr, lq) owns row m = lr and the 4 CONSECUTIVE columns lq*4..+3 -> one 16-byte (f32) or
  // 8-byte (f16) store per fragment instead of four scattered scalar stores.  Same contract as gemm_kernel: bias,
  // activation, residual at the (row-mapped) destination row.
  // The launcher guarantees N, ldc, ldr multiples of 4 (16-byte rows), so every store below is one vector.
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int row = m0 + wm * WTM + i * MF + lr;
    if (row >= p.M) continue;
    int drow = p.rowmap ? p.rowmap[row] : row;
    if (drow < 0) continue;
    int rrow = p.res_mod > 0 ? drow % p.res_mod : drow;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      int col = n0 + wn * WTN + j * MF + lq * 4;
      if (col >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
      if (p.bias) {
        float4 bv = *(const float4*)(p.bias + col);
        v[0] += bv.x, v[1] += bv.y, v[2] += bv.z, v[3] += bv.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = p.act == ACT_GELU ? gelu_half_gemm(v[r], p.out_f16) : apply_act(v[r], p.act);
      if (p.res) {
        float4 rv = *(const float4*)(p.res + (long)rrow * p.ldr + col);
        v[0] += rv.x, v[1] += rv.y, v[2] += rv.z, v[3] += rv.w;
      }
      if (p.out_f16 == 2) {        // x3 row (common.h): hi / lo halves of the 4 columns, inside one 32-block
        h4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          half_t a, b;
          split_f16(v[r], a, b);
          hi[r] = a, lo[r] = b;
        }
        half_t* cp = (half_t*)p.C + (long)drow * p.ldc + x3_col(col);
        *(h4*)cp = hi;
        *(h4*)(cp + 32) = lo;
      } else if (p.out_f16) {
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
}

int gemm_f16_p8_launch(const GemmP& p, hipStream_t s);   // gemm_f16_p8.hip

// returns SAMPT_ERR_UNSUPPORTED when the shape does not fit this kernel (caller falls back to gemm_kernel)
int gemm_f16_glds_launch(const GemmP& p, hipStream_t s) {
  {
    int rc = gemm_f16_p8_launch(p, s);
    if (rc != SAMPT_ERR_UNSUPPORTED) return rc;
  }
  if (p.conv || p.w_kn || p.nb1 * p.nb2 != 1 || p.K % 64 || p.M < 128 || p.N < 128) return SAMPT_ERR_UNSUPPORTED;
  if ((p.lda % 8) || (p.ldw % 8) || (((uintptr_t)p.A | (uintptr_t)p.W) & 15)) return SAMPT_ERR_UNSUPPORTED;
  if ((p.N % 4) || (p.ldc % 4) || (p.res && (p.ldr % 4))) return SAMPT_ERR_UNSUPPORTED;   // vector epilogue only
  GemmP q = p;
  const dim3 block(256);
  // 128 x 160 tiles wherever N is a multiple of 160 (the ViT-H / ViT-L widths): every GEMM is then a whole number of
  // generations of the 1024 resident workgroups (+3.6 .. +4.8 % per shape, profiles/r2_v4_gemm_microbench_{default,bn160}.log)
  const bool wide = p.N % 160 == 0 && p.out_f16 != 2;
  const int nt_m = cdiv(p.M, 128), nt_n = wide ? p.N / 160 : cdiv(p.N, 128);
  int R = 8;                                        // strip height of the XCD-aware tile order (see the kernel)
  while (R > 1 && cdiv(nt_m, R) < 16) R /= 2;       // keep >= 2 strips per XCD so that all 8 XCDs get work
  q.xcd_swizzle = R;
  const dim3 grid(8 * cdiv(cdiv(nt_m, R), 8) * R * nt_n, 1, 1);
  if (wide) hipLaunchKernelGGL((gemm_f16_glds<128, 160, 64, 80>), grid, block, 0, s, q);
  else hipLaunchKernelGGL((gemm_f16_glds<128, 128, 64, 64>), grid, block, 0, s, q);
  SAMPT_CHECK_LAUNCH("gemm_f16_glds");
  return SAMPT_OK;
}

}  // namespace sampt
