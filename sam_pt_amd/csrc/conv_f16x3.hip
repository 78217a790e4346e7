// fp32-grade NHWC convolution on the fp16 matrix pipe ("f16x3") for the tracker's feature encoder (pips.py:191-287).
//
// The encoder's trajectories must stay index-identical to the fp32 reference, so its convolutions cannot simply run in
// fp16.  Instead every fp32 operand is split into two halves, x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22-23
// significant bits together), and the product is evaluated as hi*hi + hi*lo + lo*hi with three
// v_mfma_f32_16x16x32_f16 (exact fp16 products, fp32 accumulation); the dropped lo*lo term is below 2^-22 relative.  On
// random data the result is closer to the fp64 convolution than an fp32 FMA chain is (tests/test_gpu_kernels.py), at 3/16
// of the fp32 MFMA's issue cycles.
//   * weights arrive pre-split from the host: half [2][Cout][K] (hi plane, lo plane), K = KH*KW*Cin, scaled by
//     2^F16X3_WSHIFT so the lo plane stays in fp16's normal range; the epilogue multiplies by p.alpha = 2^-F16X3_WSHIFT;
//   * activations are split on the fly while they are staged into LDS (post-InstanceNorm values are O(1), |x| < 65504) —
//     or arrive PRE-SPLIT as two fp16 NHWC planes written by their producer (AHL: p.A = hi plane, p.A_lo = lo plane; the
//     tracker encoder's InstanceNorm does that).  A 3 x 3 convolution stages every input element 9 times, so splitting it
//     once where it is produced removes ~16 VALU operations per staged float4 from a loop that is bound by exactly those;
//   * requires Cin % 32 == 0, so a 32-deep K slab never straddles a filter tap: no div/mod in the main loop.
//
// Tile: 128 x BN (BN = 64 / 96 / 128 = the encoder's channel counts) x 32, 4 waves (2 x 2), LDS double-buffered with one
// barrier per slab; MFMAs are issued with swapped operands so each lane owns 4 consecutive output channels (16-byte
// stores).
#include "common.h"

namespace sampt {

// PF = prefetch distance in K slabs: the global loads of slab kt + PF are issued while slab kt is multiplied (PF register
// sets; the split into fp16 planes happens when a set is stored to LDS, one slab ahead of its use).
template <int BM, int BN, int PF>
__global__ __launch_bounds__(256) void k_conv_f16x3(GemmP p) {
  constexpr int BK = 32, LDH = BK + 8;
  constexpr int WTM = BM / 2, WTN = BN / 2, FM = WTM / 16, FN = WTN / 16;
  constexpr int A_IT = BM * 8 / 256;                   // float4 (4 k) vectors per thread and slab
  constexpr int B_VEC = BN * 4, B_IT = (B_VEC + 255) / 256;   // h8 (8 k) vectors per plane
  static_assert(WTN % 16 == 0 && WTM % 16 == 0, "wave tile must be made of 16x16 fragments");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // 2 * (2 BM + 2 BN) * LDH halves (<= 80 KiB)
  typedef half_t (*TileA)[BM][LDH];
  typedef half_t (*TileB)[BN][LDH];
  TileA Ah = (TileA)smem_raw, Al = Ah + 2;
  TileB Bh = (TileB)(Al + 2), Bl = Bh + 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // 1-D grid, the N tiles of one M tile adjacent in dispatch order (they share the activation tile through L2)
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (int)(blockIdx.x / ntn) * BM, n0 = (int)(blockIdx.x % ntn) * BN;
  const float* __restrict__ A = (const float*)p.A;
  const half_t* __restrict__ Wh = (const half_t*)p.W;
  const half_t* __restrict__ Wl = (const half_t*)p.W_lo;

  int a_row[A_IT], a_kv[A_IT], a_iy0[A_IT], a_ix0[A_IT];
  long a_off[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int v = tid + i * 256;
    a_row[i] = v >> 3, a_kv[i] = (v & 7) * 4;
    const int m = m0 + a_row[i];
    a_ok[i] = m < p.M;
    const int ohw = p.OH * p.OW;
    const int img = m / ohw, rem = m - img * ohw;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    a_off[i] = (long)img * p.cH * p.cW * p.cC;
    a_iy0[i] = oy * p.cstride - p.cpad;
    a_ix0[i] = ox * p.cstride - p.cpad;
  }
  // filter tap / channel offset of the slab being LOADED (uniform over the workgroup)
  int l_ky = 0, l_kx = 0, l_ci = 0, l_k = 0;

  float4 ra0[A_IT], ra1[PF > 1 ? A_IT : 1];
  h8 rbh0[B_IT], rbl0[B_IT], rbh1[PF > 1 ? B_IT : 1], rbl1[PF > 1 ? B_IT : 1];
  auto load_slab = [&](float4* ra, h8* rbh, h8* rbl) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = a_iy0[i] + l_ky, ix = a_ix0[i] + l_kx;
      const bool ok = a_ok[i] && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
      // branch-free: out-of-image taps read the tensor's first pixel and are zeroed afterwards (a branch per vector
      // makes hipcc fence every load)
      const float4 v = *(const float4*)(A + (ok ? a_off[i] + ((long)iy * p.cW + ix) * p.cC + l_ci + a_kv[i] : 0));
      ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int v = tid + i * 256;
      const int n = n0 + (v >> 2), k = l_k + (v & 3) * 8;
      const bool ok = (B_VEC % 256 == 0 || v < B_VEC) && n < p.N;
      const long off = ok ? (long)n * p.ldw + k : 0;
      const h8 vh = *(const h8*)(Wh + off), vl = *(const h8*)(Wl + off);
      rbh[i] = ok ? vh : (h8){0, 0, 0, 0, 0, 0, 0, 0};
      rbl[i] = ok ? vl : (h8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    l_k += BK, l_ci += BK;
    if (l_ci == p.cC) {
      l_ci = 0;
      if (++l_kx == p.KW) l_kx = 0, ++l_ky;
    }
  };
  auto store_slab = [&](int buf, const float4* ra, const h8* rbh, const h8* rbl) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const float4 v = ra[i];
      // saturating split (common.h): an activation beyond the fp16 range keeps hi finite and lo carries the rest (the ViT
      // neck reads the raw residual stream, which trained checkpoints push to O(10^2 - 10^3) in a few channels)
      const float vv[4] = {v.x, v.y, v.z, v.w};
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        half_t a, b;
        split_f16(vv[e], a, b);
        hi[e] = a, lo[e] = b;
      }
      *(h4*)&Ah[buf][a_row[i]][a_kv[i]] = hi;
      *(h4*)&Al[buf][a_row[i]][a_kv[i]] = lo;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int v = tid + i * 256;
      if (B_VEC % 256 == 0 || v < B_VEC) {
        *(h8*)&Bh[buf][v >> 2][(v & 3) * 8] = rbh[i];
        *(h8*)&Bl[buf][v >> 2][(v & 3) * 8] = rbl[i];
      }
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  const int lr = lane & 15, lq = lane >> 4;
  // slab s travels in register set s % PF; iteration kt stores slab kt + 1 to LDS buffer (kt + 1) & 1 (last read before the
  // previous barrier) and re-uses its registers for slab kt + 1 + PF
  load_slab(ra0, rbh0, rbl0);
  if (PF > 1 && nk > 1) load_slab(ra1, rbh1, rbl1);
  store_slab(0, ra0, rbh0, rbl0);
  if (nk > PF) load_slab(ra0, rbh0, rbl0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      if (PF == 1 || (kt & 1)) {            // slab kt + 1 is even (or the only set): set 0
        store_slab(cur ^ 1, ra0, rbh0, rbl0);
        if (kt + 1 + PF < nk) load_slab(ra0, rbh0, rbl0);
      } else {
        store_slab(cur ^ 1, ra1, rbh1, rbl1);
        if (kt + 1 + PF < nk) load_slab(ra1, rbh1, rbl1);
      }
    }
    h8 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      ah[i] = *(const h8*)&Ah[cur][wm * WTM + i * 16 + lr][lq * 8];
      al[i] = *(const h8*)&Al[cur][wm * WTM + i * 16 + lr][lq * 8];
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      bh[j] = *(const h8*)&Bh[cur][wn * WTN + j * 16 + lr][lq * 8];
      bl[j] = *(const h8*)&Bl[cur][wn * WTN + j * 16 + lr][lq * 8];
    }
    // operands swapped: the fragment is C^T, lane (lr, lq) holds channels lq*4 .. lq*4+3 of output pixel lr
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
    __syncthreads();
  }

  // ---- epilogue: 4 consecutive channels per lane
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = m0 + wm * WTM + i * 16 + lr;
    if (row >= p.M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * WTN + j * 16 + lq * 4;
      if (col >= p.N) continue;
      float4 v = make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha,
                             acc[i][j][3] * p.alpha);
      long orow = row;       // destination pixel row and channel (pixel shuffle of a transposed convolution, see GemmP)
      int ocol = col;
      if (p.shuf_g) {
        const int z = col / p.shuf_n, g = p.shuf_g, P = g * g;
        ocol = col - z * p.shuf_n;
        const int f = row / P, rem = row - f * P, y = rem / g, x = rem - y * g;
        orow = (long)f * 4 * P + (long)(2 * y + (z >> 1)) * 2 * g + 2 * x + (z & 1);
      }
      if (p.bias) {
        const float4 b = *(const float4*)(p.bias + ocol);
        v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
      }
      v.x = apply_act(v.x, p.act), v.y = apply_act(v.y, p.act), v.z = apply_act(v.z, p.act), v.w = apply_act(v.w, p.act);
      if (p.res) {
        const long rrow = p.res_mod > 0 ? orow % p.res_mod : orow;    // broadcast residual (positional embedding)
        const float4 r = *(const float4*)(p.res + rrow * p.ldr + ocol);
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
      }
      *(float4*)((float*)p.C + orow * p.ldc + ocol) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same convolution for PRE-SPLIT activations (p.A = hi plane, p.A_lo = lo plane: fp16 NHWC, written by the producing
// InstanceNorm): with ready-made halves in memory all four operand planes go HBM -> LDS by LDS-DMA
// (`global_load_lds_dwordx4`, no VGPR staging, no ds_write pass, no split arithmetic), the slab loop is the single-buffer
// "issue, wait, barrier, multiply" of gemm_f16.hip and the kernel is light enough (24 - 32 KiB of LDS, <= 128 VGPRs) for four
// workgroups per CU to hide each other's DMA latency.  128 x BN x 32 tile, 4 waves (2 x 2), K slab = 32 channels of one
// filter tap (Cin % 32 == 0).  LDS rows are 64 B (one slab row of one plane); a DMA instruction lands 16 rows; the image of
// an instruction is linear (M0 base + lane * 16), so the bank swizzle sits on the SOURCE side: the lane that fills LDS chunk
// position q of row r fetches source chunk q ^ F[(r >> 2) & 3], F = {0, 3, 2, 1}, and fragment reads apply the same XOR
// (every ds_read_b128 lane group then touches 16 distinct 16-byte slots).  Taps outside the image read a page of zeros.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(64))) half_t g_conv_zero_page[64];

template <int BN>
__global__ __launch_bounds__(256, BN >= 128 ? 3 : 4) void k_conv_f16x3_dma(GemmP p) {
  constexpr int BM = 128, BK = 32, ROWB = BK * 2;            // bytes per LDS row
  constexpr int WTM = 64, WTN = BN / 2, FM = WTM / 16, FN = WTN / 16;
  constexpr int NBP = 2 * BN / 16;                           // B pieces (16 rows each) over both planes
  static_assert(NBP % 4 == 0, "B pieces must divide over the 4 waves");
  __shared__ __attribute__((aligned(1024))) char lds[(2 * BM + 2 * BN) * ROWB];   // [A hi | A lo | B hi | B lo]
  constexpr int OFF_AL = BM * ROWB, OFF_BH = 2 * BM * ROWB, OFF_BL = OFF_BH + BN * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (int)(blockIdx.x / ntn) * BM, n0 = (int)(blockIdx.x % ntn) * BN;
  const char* __restrict__ Ah = (const char*)p.A;
  const char* __restrict__ Al = (const char*)p.A_lo;
  const char* __restrict__ Wh = (const char*)p.W;
  const char* __restrict__ Wl = (const char*)p.W_lo;
  const char* zero = (const char*)g_conv_zero_page;

  // DMA roles: lane l of an instruction fills row (l >> 2), chunk position (l & 3) of its 16-row piece
  const int prow = lane >> 2;
  const int fsw = (4 - (lane >> 4)) & 3;                     // F[(row >> 2) & 3] with (row >> 2) & 3 == lane >> 4
  const int csrc = ((lane & 3) ^ fsw) * 16;                  // byte offset of the source chunk inside the 64-byte slab row
  // A: this wave fills pieces 2*wave, 2*wave + 1 of both planes -> two tile rows per lane
  long a_base[2];
  int a_iy0[2], a_ix0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (2 * wave + i) * 16 + prow;
    const int ohw = p.OH * p.OW;
    const int mm = m < p.M ? m : p.M - 1;
    const int img = mm / ohw, rem = mm - img * ohw;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    a_base[i] = (long)img * p.cH * p.cW * p.cC;
    a_iy0[i] = m < p.M ? oy * p.cstride - p.cpad : -(1 << 28);     // rows beyond M: never inside the image
    a_ix0[i] = ox * p.cstride - p.cpad;
  }
  // B: pieces q = wave + 4 j over [hi plane pieces | lo plane pieces]
  long b_off[NBP / 4];
#pragma unroll
  for (int j = 0; j < NBP / 4; ++j) {
    const int q = wave + 4 * j, piece = q % (BN / 16);
    int n = n0 + piece * 16 + prow;
    if (n > p.N - 1) n = p.N - 1;
    b_off[j] = ((long)n * p.ldw) * 2 + csrc;
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lq = lane >> 4;
  const int rsw = ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) * 16);  // swizzled chunk of this lane's fragment rows
  const int a_rd = (wm * WTM + lr) * ROWB + rsw, b_rd = OFF_BH + (wn * WTN + lr) * ROWB + rsw;

  const int nk = p.K / BK;
  int l_ky = 0, l_kx = 0, l_ci = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt > 0) __syncthreads();                               // everyone done reading the previous slab
    // ---- issue the slab: A rows (tap l_ky, l_kx; channels l_ci .. +31), W rows (k = kt * 32 .. +31)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int iy = a_iy0[i] + l_ky, ix = a_ix0[i] + l_kx;
      const bool ok = iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
      const long off = (a_base[i] + ((long)iy * p.cW + ix) * p.cC + l_ci) * 2 + csrc;
      const char* sh = ok ? Ah + off : zero + csrc;
      const char* sl = ok ? Al + off : zero + csrc;
      char* dst = lds + (2 * wave + i) * 16 * ROWB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sh, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sl, (__attribute__((address_space(3))) void*)(dst + OFF_AL), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NBP / 4; ++j) {
      const int q = wave + 4 * j;                                // uniform
      const bool lo = q >= BN / 16;
      const int piece = lo ? q - BN / 16 : q;
      const char* src = (lo ? Wl : Wh) + b_off[j] + (long)kt * (BK * 2);
      char* dst = lds + (lo ? OFF_BL : OFF_BH) + piece * 16 * ROWB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    l_ci += BK;
    if (l_ci == p.cC) {
      l_ci = 0;
      if (++l_kx == p.KW) l_kx = 0, ++l_ky;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- multiply: hi*lo + lo*hi + hi*hi per fragment (operands swapped: the fragment is C^T, see the epilogue)
    h8 ah[FM], al[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      ah[i] = *(const h8*)(lds + a_rd + i * 16 * ROWB);
      al[i] = *(const h8*)(lds + OFF_AL + a_rd + i * 16 * ROWB);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const h8 bh = *(const h8*)(lds + b_rd + j * 16 * ROWB);
      const h8 bl = *(const h8*)(lds + (OFF_BL - OFF_BH) + b_rd + j * 16 * ROWB);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane (lr, lq) holds channels lq*4 .. lq*4+3 of output pixel lr of each fragment
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = m0 + wm * WTM + i * 16 + lr;
    if (row >= p.M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * WTN + j * 16 + lq * 4;
      if (col >= p.N) continue;
      float4 v = make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
      if (p.bias) {
        const float4 b = *(const float4*)(p.bias + col);
        v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
      }
      v.x = apply_act(v.x, p.act), v.y = apply_act(v.y, p.act), v.z = apply_act(v.z, p.act), v.w = apply_act(v.w, p.act);
      if (p.res) {
        const long rrow = p.res_mod > 0 ? row % p.res_mod : row;
        const float4 r = *(const float4*)(p.res + rrow * p.ldr + col);
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
      }
      *(float4*)((float*)p.C + (long)row * p.ldc + col) = v;
    }
  }
}

int conv_f16x3(const GemmP& p_in, hipStream_t s) {
  GemmP p = p_in;
  if (!p.A || !p.W || !p.W_lo || !p.C || p.M <= 0 || p.N <= 0) return SAMPT_ERR_ARG;
  if (!p.conv || p.cC % 32 || p.K != p.KH * p.KW * p.cC || p.ldw % 8 || p.N % 4 || p.ldc % 4) return SAMPT_ERR_UNSUPPORTED;
  if (p.shuf_g && (p.shuf_n <= 0 || p.shuf_n % 4 || p.N != 4 * p.shuf_n || p.M % (p.shuf_g * p.shuf_g))) return SAMPT_ERR_ARG;
  if (p.cpadw >= 0 || p.rowmap || p.a_rowmap || p.nb1 * p.nb2 != 1 || (p.res && p.ldr % 4)) return SAMPT_ERR_UNSUPPORTED;
  if (((uintptr_t)p.A | (uintptr_t)p.A_lo | (uintptr_t)p.W | (uintptr_t)p.W_lo | (uintptr_t)p.C | (uintptr_t)p.bias |
       (uintptr_t)p.res) & 15)
    return SAMPT_ERR_ARG;
  // (Round 6 tried a kernel of its own for the tall, short-K 1 x 1 case — the mask decoder's image-side projections, M = 98 304,
  //  K = 256, N = 256 .. 512: 128 rows x all N per workgroup, A straight into operand registers, W through a 4-slot LDS-DMA ring —
  //  and measured 99.7 / 132.6 / 169.6 us against 98.6 / 152.2 / 190.1 us here, 208.1 vs 209.0 ms per clip: both sit at ~2 TB/s of
  //  A + C because every workgroup re-fetches W and a CU moves ~25 GB/s through LDS-DMA whatever the source; not kept.
  //  profiles/r6_c9_gemm_x3_rows_vs_register_staged.log)
  if (g_conv_halo && conv3x3_halo_eligible(p)) return conv3x3_halo_x3(p, s);
  if (g_gemm_x3_wres && gemm_x3_wres_eligible(p)) return gemm_x3_wres(p, s);
  if (p.A_lo) {   // pre-split activations: the LDS-DMA kernel
    if (p.shuf_g) return SAMPT_ERR_UNSUPPORTED;
    const int BNd = p.N <= 64 ? 64 : (p.N <= 96 ? 96 : 128);
    dim3 gridd((unsigned)((long)cdiv(p.N, BNd) * cdiv(p.M, 128))), blockd(256);
    if (BNd == 64) hipLaunchKernelGGL(k_conv_f16x3_dma<64>, gridd, blockd, 0, s, p);
    else if (BNd == 96) hipLaunchKernelGGL(k_conv_f16x3_dma<96>, gridd, blockd, 0, s, p);
    else hipLaunchKernelGGL(k_conv_f16x3_dma<128>, gridd, blockd, 0, s, p);
    SAMPT_CHECK_LAUNCH("conv_f16x3_dma");
    return SAMPT_OK;
  }
  const int BN = p.N <= 32 ? 32 : (p.N <= 64 ? 64 : (p.N <= 96 ? 96 : 128));
  const size_t lds = (size_t)2 * (2 * 128 + 2 * BN) * (32 + 8) * sizeof(half_t);
  // (A single LDS buffer — 40 KiB, four workgroups per CU instead of two, two barriers per slab — was measured in round 4 and is
  // within noise: decode chain 24.0 vs 23.6 ms, 113.3 vs 112.9 fps, profiles/r4_c9_*.  These launches are not occupancy-bound.)
  // One K slab of register prefetch.  Two slabs in flight (a second register set) measured slower everywhere — decode chain
  // 24.70 vs 24.34 ms, tracker-encoder pass 6.54 vs 6.08 ms (profiles/r2_v16_*): these kernels are not bound by the latency of
  // their global loads, and the extra 30 registers cost the 64- and 96-column tiles a resident wave.
  static bool raised = false;
  if (!raised) {  // 128 x 128 tiles need 80 KiB, above the default 64 KiB dynamic-LDS limit (gfx950: 160 KiB per CU)
    const void* fns[] = {(const void*)k_conv_f16x3<128, 64, 1>, (const void*)k_conv_f16x3<128, 96, 1>,
                         (const void*)k_conv_f16x3<128, 128, 1>};
    for (const void* f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return SAMPT_ERR_HIP;
    raised = true;
  }
  dim3 grid((unsigned)((long)cdiv(p.N, BN) * cdiv(p.M, 128))), block(256);
#define CONV_LAUNCH(BNv) hipLaunchKernelGGL((k_conv_f16x3<128, BNv, 1>), grid, block, lds, s, p)
  if (BN == 32) CONV_LAUNCH(32);   // (51 KiB: under the default limit)
  else if (BN == 64) CONV_LAUNCH(64);
  else if (BN == 96) CONV_LAUNCH(96);
  else CONV_LAUNCH(128);
#undef CONV_LAUNCH
  SAMPT_CHECK_LAUNCH("conv_f16x3");
  return SAMPT_OK;
}

}  // namespace sampt
