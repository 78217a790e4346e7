// MFMA GEMM + implicit-GEMM (NHWC) convolution for gfx950.
//
//  * f32 path: v_mfma_f32_16x16x4_f32 — exact fp32 (bitwise a k-ordered fmaf chain), used for the whole PIPS
//    tracker (trajectories must stay index-identical to the fp32 reference), the mask decoder and the
//    "exact" ViT mode.  Peak 157 TFLOP/s.
//  * f16 path: v_mfma_f32_16x16x32_f16 — fp16 operands, fp32 accumulate, used for the ViT image encoder
//    (qkv / proj / MLP / patch-embed / neck GEMMs).  Peak ~2.5 PFLOP/s.
//
// Structure: 256 threads = 4 waves (2 x 2), block tile BM x BN (128x128 or 64x64), wave tile (BM/2)x(BN/2) made of
// 16x16 MFMA fragments; K streamed through LDS in BK slabs with register prefetch of the next slab while the
// current one is multiplied.  Operands are K-contiguous ("A row-major, W = torch Linear weight [N][K]"), so both
// fragment reads are contiguous 4-byte (f32) / 16-byte (f16) LDS reads.  The optional W_KN layout (W stored
// [K][N], f32 only) serves P.V products.  The A operand may be an im2col view of an NHWC tensor computed on the
// fly (convolution = GEMM with M = output pixels, K = (ky,kx,ci), N = Cout).
//
// Latency-bound shapes of the decoder / mixer get two extra paths (both deterministic):
//  * split-K: grids with too few tiles to fill 256 CUs slice K over blockIdx.z, write fp32 partial tiles to a
//    caller-provided workspace and a second kernel reduces them in a fixed order and applies the epilogue;
//  * skinny (M <= 32): no MFMA at all — each wave owns two output columns, lanes stride K with 16-byte loads of the
//    weight row (weight-bandwidth bound, one pass over W), wave-shuffle reduction.
#include <stdlib.h>

#include "common.h"

namespace sampt {

template <typename T> struct GT;
template <> struct GT<float> {
  static constexpr int VEC = 4, PAD = 4;
  typedef float4 vec_t;
  __device__ static vec_t zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct GT<half_t> {
  static constexpr int VEC = 8, PAD = 8;
  typedef h8 vec_t;
  __device__ static vec_t zero() { return (h8){0, 0, 0, 0, 0, 0, 0, 0}; }
};

template <typename T, int BM, int BN, int BK, bool CONV, bool W_KN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  typedef GT<T> G;
  typedef typename G::vec_t vec_t;
  constexpr int VEC = G::VEC, PAD = G::PAD;
  constexpr int WTM = BM / 2, WTN = BN / 2, FM = WTM / 16, FN = WTN / 16;
  constexpr int KV = BK / VEC;                  // vectors per A/W row of the slab
  constexpr int A_IT = BM * KV / 256;
  constexpr int B_IT = W_KN ? (BK * (BN / VEC) / 256) : (BN * KV / 256);
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for 256 threads");
  static_assert(!W_KN || sizeof(T) == 4, "W_KN layout is f32 only");

  constexpr int B_ROWS = W_KN ? BK : BN;
  constexpr int B_COLS = W_KN ? BN + PAD : BK + PAD;
  __shared__ __attribute__((aligned(16))) T As[BM][BK + PAD];
  __shared__ __attribute__((aligned(16))) T Bs[B_ROWS][B_COLS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  int z = blockIdx.z, ks = 0;
  if (p.splitk > 1) {
    ks = z % p.splitk;
    z /= p.splitk;
  }
  const int z1 = z / p.nb2, z2 = z - z1 * p.nb2;
  // K range of this block (split-K slices are multiples of BK)
  const int kslice = p.splitk > 1 ? ((p.K + p.splitk * BK - 1) / (p.splitk * BK)) * BK : p.K;
  const int kbeg = ks * kslice;
  const int kend = min(p.K, kbeg + kslice);

  const T* __restrict__ A = (const T*)p.A + z1 * p.sA1 + z2 * p.sA2;
  const T* __restrict__ W = (const T*)p.W + z1 * p.sW1 + z2 * p.sW2;

  // ---- per-thread A row bookkeeping (rows are fixed across the K loop)
  int a_row[A_IT], a_kv[A_IT];
  long a_off[A_IT];          // plain: m*lda ; conv: image base offset
  int a_iy0[A_IT], a_ix0[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int v = tid + i * 256;
    a_row[i] = v / KV;
    a_kv[i] = (v % KV) * VEC;
    int m = m0 + a_row[i];
    a_ok[i] = m < p.M;
    if (CONV) {
      int ohw = p.OH * p.OW;
      int img = m / ohw, rem = m - img * ohw;
      int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_off[i] = (long)img * p.cH * p.cW * p.cC;
      a_iy0[i] = oy * p.cstride - p.cpad;
      a_ix0[i] = ox * p.cstride - (p.cpadw >= 0 ? p.cpadw : p.cpad);
    } else {
      a_off[i] = (long)(p.a_rowmap && a_ok[i] ? p.a_rowmap[m] : m) * p.lda;
      a_iy0[i] = a_ix0[i] = 0;
    }
  }

  auto load_a = [&](int i, int k0) -> vec_t {
    int k = k0 + a_kv[i];
    if (!a_ok[i] || k >= kend) return G::zero();
    if (CONV) {
      int kpos = k / p.cC, ci = k - kpos * p.cC;
      int ky = kpos / p.KW, kx = kpos - ky * p.KW;
      int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      if (iy < 0 || iy >= p.cH || ix < 0 || ix >= p.cW) return G::zero();
      return *(const vec_t*)(A + a_off[i] + ((long)iy * p.cW + ix) * p.cC + ci);
    } else {
      return *(const vec_t*)(A + a_off[i] + k);
    }
  };
  auto load_b = [&](int i, int k0) -> vec_t {
    int v = tid + i * 256;
    if (W_KN) {
      constexpr int NV = BN / VEC;
      int kr = v / NV, nv = (v % NV) * VEC;
      int k = k0 + kr, n = n0 + nv;
      if (k >= kend || n >= p.N) return G::zero();
      return *(const vec_t*)(W + (long)k * p.ldw + n);
    } else {
      int r = v / KV, kv = (v % KV) * VEC;
      int n = n0 + r, k = k0 + kv;
      if (n >= p.N || k >= kend) return G::zero();
      return *(const vec_t*)(W + (long)n * p.ldw + k);
    }
  };
  auto store_a = [&](int i, const vec_t& v) { *(vec_t*)&As[a_row[i]][a_kv[i]] = v; };
  auto store_b = [&](int i, const vec_t& val) {
    int v = tid + i * 256;
    if (W_KN) {
      constexpr int NV = BN / VEC;
      *(vec_t*)&Bs[v / NV][(v % NV) * VEC] = val;
    } else {
      *(vec_t*)&Bs[v / KV][(v % KV) * VEC] = val;
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  vec_t ra[A_IT], rb[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) ra[i] = load_a(i, kbeg);
#pragma unroll
  for (int i = 0; i < B_IT; ++i) rb[i] = load_b(i, kbeg);

  const int nk = (kend - kbeg + BK - 1) / BK;
  const int lr = lane & 15, lq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) store_a(i, ra[i]);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) store_b(i, rb[i]);
    __syncthreads();
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[i] = load_a(i, kbeg + (kt + 1) * BK);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) rb[i] = load_b(i, kbeg + (kt + 1) * BK);
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float a[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = ((const float(*)[BK + PAD])As)[wm * WTM + i * 16 + lr][kk * 4 + lq];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (W_KN) b[j] = ((const float(*)[B_COLS])Bs)[kk * 4 + lq][wn * WTN + j * 16 + lr];
          else b[j] = ((const float(*)[B_COLS])Bs)[wn * WTN + j * 16 + lr][kk * 4 + lq];
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // plain: BK / 32 consecutive 32-deep steps; x3 (GemmP::x3, BK == 64): lo.hi, hi.lo, hi.hi of the slab's 32 real k
      const int nterm = p.x3 ? 3 : BK / 32;
      for (int kk = 0; kk < nterm; ++kk) {
        const int ka = p.x3 ? (kk == 0 ? 1 : 0) : kk, kb = p.x3 ? (kk == 1 ? 1 : 0) : kk;
        h8 a[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = *(const h8*)&As[wm * WTM + i * 16 + lr][ka * 32 + lq * 8];
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = *(const h8*)&Bs[wn * WTN + j * 16 + lr][kb * 32 + lq * 8];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- split-K: raw partial tile, the epilogue runs in k_splitk_reduce
  if (p.splitk > 1) {
    float* part = p.splitk_ws + (long)ks * p.M * p.N;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = m0 + wm * WTM + i * 16 + lq * 4 + r;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          int col = n0 + wn * WTN + j * 16 + lr;
          if (col < p.N) part[(long)row * p.N + col] = acc[i][j][r];
        }
      }
    return;
  }

  // ---- epilogue: C/D fragment layout col = lane & 15, row = (lane >> 4) * 4 + reg
  const int* rowmap = p.rowmap ? p.rowmap + z1 * p.sRowmap1 : nullptr;
  const long c_off = z1 * p.sC1 + z2 * p.sC2;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = m0 + wm * WTM + i * 16 + lq * 4 + r;
      if (row >= p.M) continue;
      int drow = rowmap ? rowmap[row] : row;
      if (drow < 0) continue;
      int rrow = p.res_mod > 0 ? drow % p.res_mod : drow;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        int col = n0 + wn * WTN + j * 16 + lr;
        if (col >= p.N) continue;
        float v = acc[i][j][r] * p.alpha;
        if (p.bias) v += p.bias[col];
        v = (sizeof(T) == 2 && p.act == ACT_GELU) ? gelu_half_gemm(v, p.out_f16) : apply_act(v, p.act);
        if (p.res) v += p.res[(long)rrow * p.ldr + col];
        if (sizeof(T) == 2 && p.out_f16 == 2) {
          half_t hi, lo;
          split_f16(v, hi, lo);
          half_t* cp = (half_t*)p.C + c_off + (long)drow * p.ldc + x3_col(col);
          cp[0] = hi, cp[32] = lo;
        } else if (sizeof(T) == 2 && p.out_f16) ((half_t*)p.C)[c_off + (long)drow * p.ldc + col] = (half_t)v;
        else ((float*)p.C)[c_off + (long)drow * p.ldc + col] = v;
      }
    }
  }
}

// fixed-order reduction of the split-K partials + epilogue (f32 output only)
__global__ void k_splitk_reduce(GemmP p) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)p.M * p.N;
  if (i >= total) return;
  int row = (int)(i / p.N), col = (int)(i - (long)row * p.N);
  float v = 0.f;
  for (int s = 0; s < p.splitk; ++s) v += p.splitk_ws[(long)s * total + i];
  v *= p.alpha;
  if (p.bias) v += p.bias[col];
  v = apply_act(v, p.act);
  int rrow = p.res_mod > 0 ? row % p.res_mod : row;
  if (p.res) v += p.res[(long)rrow * p.ldr + col];
  ((float*)p.C)[(long)row * p.ldc + col] = v;
}

// ---------------------------------------------------------------------------------------------
// skinny f32 GEMM (M <= MT): wave -> 2 output columns, lanes stride K (float4), shuffle reduce
// ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_f32(GemmP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * 2;
  if (n0 >= p.N) return;
  const bool two = n0 + 1 < p.N;
  const float* A = (const float*)p.A;
  const float* W0 = (const float*)p.W + (long)n0 * p.ldw;
  const float* W1 = two ? W0 + p.ldw : W0;
  float acc0[MT], acc1[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc0[m] = acc1[m] = 0.f;
  for (int k = lane * 4; k < p.K; k += 256) {
    float4 w0 = *(const float4*)(W0 + k), w1 = *(const float4*)(W1 + k);
    // branch-free: rows beyond M re-read row M-1 (a per-row branch makes hipcc serialise the loads behind vmcnt(0))
    float4 a[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = *(const float4*)(A + (long)(m < p.M ? m : p.M - 1) * p.lda + k);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc0[m] += a[m].x * w0.x + a[m].y * w0.y + a[m].z * w0.z + a[m].w * w0.w;
      acc1[m] += a[m].x * w1.x + a[m].y * w1.y + a[m].z * w1.z + a[m].w * w1.w;
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (m < p.M) {
      float s0 = wave_sum(acc0[m]), s1 = wave_sum(acc1[m]);
      if (lane < 2 && (lane == 0 || two)) {
        int col = n0 + lane;
        float v = (lane == 0 ? s0 : s1) * p.alpha;
        if (p.bias) v += p.bias[col];
        v = apply_act(v, p.act);
        int rrow = p.res_mod > 0 ? m % p.res_mod : m;
        if (p.res) v += p.res[(long)rrow * p.ldr + col];
        ((float*)p.C)[(long)m * p.ldc + col] = v;
      }
    }
  }
}

int gemm_f16_glds_launch(const GemmP& p, hipStream_t s);  // gemm_f16.hip

// tile / split-K selection shared by the dispatcher and gemm_f32_plan_splitk
static int plan_tiles_splitk(const GemmP& p, bool plain, int& BM_out, bool& big_out) {
  const int batch = p.nb1 * p.nb2;
  const long big_tiles = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * batch;
  const bool big = p.M > 64 && p.N > 64 && big_tiles >= 192;
  const int BM = big ? 128 : 64;
  const long tiles = (long)cdiv(p.M, BM) * cdiv(p.N, BM) * batch;
  BM_out = BM, big_out = big;
  if (plain && !p.out_f16 && !p.x3 && p.splitk_ws && tiles < 128 && p.K >= 512) {
    int want = (int)((255 + tiles) / tiles);
    int maxs = p.K / 256;
    int sk = want < maxs ? want : maxs;
    if (sk > 16) sk = 16;
    while (sk > 1 && (size_t)sk * p.M * p.N > p.splitk_ws_floats) --sk;
    if (sk > 1) return sk;
  }
  return 1;
}

static bool thin_f32_eligible(const GemmP& p, bool plain);

int gemm_f32_plan_splitk(const GemmP& p) {
  const bool plain = !p.conv && !p.w_kn && p.nb1 * p.nb2 == 1 && !p.rowmap && !p.a_rowmap;
  if (plain && p.M <= 16) return 1;  // skinny path
  if (thin_f32_eligible(p, plain)) return 1;             // thin path: K is split inside the workgroup
  int BM;
  bool big;
  return plan_tiles_splitk(p, plain, BM, big);
}

// ---------------------------------------------------------------------------------------------
// thin f32 GEMM for the latency-bound token-side products (PIPS / CoTracker mixers: 64 .. 400 rows; decoder tokens):
// a workgroup owns a (16 * FM) x 16 output tile and its NWV WAVES SPLIT K; the partial tiles meet in LDS and are summed
// in wave order (deterministic), so a 64 x 512 x 2048 product is 128 workgroups and ONE launch — where the 64 x 64 tile
// needed a split-K grid plus a reduction launch.  Exact fp32: v_mfma_f32_16x16x4_f32.  These products are bound by the
// latency of their (L2 / Infinity-Cache resident) operands, not by bandwidth or FLOPs: NWV is chosen so that a wave's
// share of K is at most 8 chunks of 16 and ALL its loads are issued before the first MFMA — one memory round trip per
// workgroup (16 waves for K = 2048, 4 for K = 512).
// No LDS staging: a lane loads 16 bytes of its A row / W row per 16-deep chunk straight into the MFMA operands (the K
// index a lane supplies to MFMA j of a chunk is chunk*16 + 4*(lane/16) + j for both operands, so every k is multiplied
// exactly once).  Swapped MFMA operands give C^T fragments: lane (lr, lq) owns row lr and 4 consecutive columns.
// ---------------------------------------------------------------------------------------------
template <int FM, int NWV>
__global__ __launch_bounds__(NWV * 64) void gemm_thin_f32(GemmP p) {
  __shared__ f32x4 red[NWV][FM][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * FM);
  const float* __restrict__ A = (const float*)p.A;
  const float* __restrict__ W = (const float*)p.W;
  const int wcol = n0 + lr < p.N ? n0 + lr : p.N - 1;
  const float* wp = W + (long)wcol * p.ldw + lq * 4;
  const float* ap[FM];
#pragma unroll
  for (int f = 0; f < FM; ++f) {
    const int r = m0 + f * 16 + lr;
    ap[f] = A + (long)(r < p.M ? r : p.M - 1) * p.lda + lq * 4;
  }
  const int nchunk = (p.K + 15) >> 4;
  const int c_lo = (nchunk * wave) / NWV, c_hi = (nchunk * (wave + 1)) / NWV;
  f32x4 acc[FM];
#pragma unroll
  for (int f = 0; f < FM; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int U = FM >= 4 ? 4 : 8;    // chunks whose loads are in flight together (one round trip when c_hi - c_lo <= U)
  for (int c = c_lo; c < c_hi; c += U) {
    float4 wv[U], av[U][FM];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (c + u) << 4;
      const bool in = c + u < c_hi && k + lq * 4 < p.K;     // K % 4 == 0: a float4 is wholly inside or outside
      const int ko = in ? k : 0;
      wv[u] = *(const float4*)(wp + ko);
#pragma unroll
      for (int f = 0; f < FM; ++f) av[u][f] = *(const float4*)(ap[f] + ko);
      if (!in) wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int f = 0; f < FM; ++f) {
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].x, av[u][f].x, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].y, av[u][f].y, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].z, av[u][f].z, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].w, av[u][f].w, acc[f], 0, 0, 0);
      }
  }
#pragma unroll
  for (int f = 0; f < FM; ++f) red[wave][f][lane] = acc[f];
  __syncthreads();
  // fragment f is finished by wave f (FM <= 4 <= NWV): sum the K shares in wave order, epilogue, store
  if (wave < FM) {
    const int f = wave;
    f32x4 v = red[0][f][lane];
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += red[w][f][lane];
    const int row = m0 + f * 16 + lr, col = n0 + lq * 4;
    if (row < p.M && col < p.N) {
      float o[4] = {v[0] * p.alpha, v[1] * p.alpha, v[2] * p.alpha, v[3] * p.alpha};
      if (p.bias) {
        const float4 b = *(const float4*)(p.bias + col);
        o[0] += b.x, o[1] += b.y, o[2] += b.z, o[3] += b.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = apply_act(o[r], p.act);
      if (p.res) {
        const int rrow = p.res_mod > 0 ? row % p.res_mod : row;
        const float4 rv = *(const float4*)(p.res + (long)rrow * p.ldr + col);
        o[0] += rv.x, o[1] += rv.y, o[2] += rv.z, o[3] += rv.w;
      }
      *(float4*)((float*)p.C + (long)row * p.ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// the thin kernel takes plain f32 GEMMs whose 64 x 64 tiling would leave most CUs idle (the split-K regime)
static bool thin_f32_eligible(const GemmP& p, bool plain) {
  if (!plain || p.out_f16 || p.M <= 16 || p.K < 64) return false;
  if ((p.N % 4) || (p.ldc % 4) || (p.res && (p.ldr % 4)) || (p.lda % 4) || (p.ldw % 4) || (p.K % 4)) return false;
  if (((uintptr_t)p.C & 15) || (p.bias && ((uintptr_t)p.bias & 15)) || (p.res && ((uintptr_t)p.res & 15))) return false;
  const long tiles64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64);
  return tiles64 < 128;
}

int g_thin_min_wgs = 256;    // experiment knob (sampt_gemm_set_thin_min_wgs): see gemm_thin_f32_launch

static int gemm_thin_f32_launch(const GemmP& p, hipStream_t s) {
  const long cols = cdiv(p.N, 16);
  int FM = 4;                                  // the tallest tile that still yields >= 256 workgroups (>= 1 per CU)
  while (FM > 1 && cols * cdiv(p.M, 16 * FM) < g_thin_min_wgs) FM >>= 1;
  const int nchunk = cdiv(p.K, 16), per_wave = FM >= 4 ? 4 : 8;
  int NWV = nchunk > 8 * per_wave ? 16 : (nchunk > 4 * per_wave ? 8 : 4);   // a wave's K share: one batch of loads
  dim3 grid((unsigned)cols, (unsigned)cdiv(p.M, 16 * FM)), block(NWV * 64);
#define THIN(FMv, NWVv) hipLaunchKernelGGL((gemm_thin_f32<FMv, NWVv>), grid, block, 0, s, p)
  if (FM == 4) { if (NWV == 16) THIN(4, 16); else if (NWV == 8) THIN(4, 8); else THIN(4, 4); }
  else if (FM == 2) { if (NWV == 16) THIN(2, 16); else if (NWV == 8) THIN(2, 8); else THIN(2, 4); }
  else { if (NWV == 16) THIN(1, 16); else if (NWV == 8) THIN(1, 8); else THIN(1, 4); }
#undef THIN
  SAMPT_CHECK_LAUNCH("gemm_thin_f32");
  return SAMPT_OK;
}

template <typename T>
static int gemm_dispatch(const GemmP& p_in, hipStream_t s) {
  constexpr int VEC = GT<T>::VEC;
  GemmP p = p_in;
  if constexpr (sizeof(T) == 2) {
    if (!p.force_generic) {
      int rc = gemm_f16_glds_launch(p, s);  // LDS-DMA kernel for the big ViT GEMMs
      if (rc != SAMPT_ERR_UNSUPPORTED) return rc;
    }
  }
  if (!p.A || !p.W || !p.C || p.M <= 0 || p.N <= 0 || p.K <= 0) return SAMPT_ERR_ARG;
  if (p.K % VEC) return SAMPT_ERR_ARG;
  if (sizeof(T) == 4 ? (p.x3 != 0 || p.out_f16 != 0) : ((p.x3 && (p.K % 64 || p.conv)) || (p.out_f16 == 2 && p.ldc % 64)))
    return SAMPT_ERR_ARG;
  if (p.conv) {
    if (p.cC % VEC || p.K != p.KH * p.KW * p.cC || p.w_kn) return SAMPT_ERR_ARG;
  } else if (p.lda % VEC) {
    return SAMPT_ERR_ARG;
  }
  if (p.w_kn ? (p.N % VEC || p.ldw % VEC) : (p.ldw % VEC)) return SAMPT_ERR_ARG;
  if (((uintptr_t)p.A | (uintptr_t)p.W) & 15) return SAMPT_ERR_ARG;
  if ((p.sA1 | p.sA2 | p.sW1 | p.sW2) % VEC) return SAMPT_ERR_ARG;
  const int batch = p.nb1 * p.nb2;
  const bool plain = !p.conv && !p.w_kn && batch == 1 && !p.rowmap && !p.a_rowmap;

  // ---- skinny path
  if constexpr (sizeof(T) == 4) {
    if (plain && p.M <= 16) {
      p.splitk = 1;
      dim3 grid(cdiv(p.N, 8)), block(256);
      if (p.M <= 4) hipLaunchKernelGGL(gemm_skinny_f32<4>, grid, block, 0, s, p);
      else hipLaunchKernelGGL(gemm_skinny_f32<16>, grid, block, 0, s, p);
      SAMPT_CHECK_LAUNCH("gemm_skinny");
      return SAMPT_OK;
    }
  }

  if constexpr (sizeof(T) == 4) {
    if (thin_f32_eligible(p, plain)) return gemm_thin_f32_launch(p, s);
  }

  // ---- tile selection: the big tile only when it still yields enough workgroups for 256 CUs;
  // ---- split-K for latency-bound shapes (few tiles, long K); deterministic two-stage reduction
  int BM;
  bool big;
  p.splitk = plan_tiles_splitk(p, plain, BM, big);
  const int BN = BM;
  dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), batch * p.splitk), block(256);
#define LAUNCH(BMv, BKv, CONVv, KNv) \
  hipLaunchKernelGGL((gemm_kernel<T, BMv, BMv, BKv, CONVv, KNv>), grid, block, 0, s, p)
  if constexpr (sizeof(T) == 4) {
    if (p.w_kn) {
      if (big) LAUNCH(128, 16, false, true); else LAUNCH(64, 16, false, true);
    } else if (p.conv) {
      if (big) LAUNCH(128, 32, true, false); else LAUNCH(64, 64, true, false);
    } else {
      if (big) LAUNCH(128, 32, false, false); else LAUNCH(64, 64, false, false);
    }
  } else {
    if (p.w_kn) return SAMPT_ERR_UNSUPPORTED;
    if (p.conv) {
      if (big) LAUNCH(128, 64, true, false); else LAUNCH(64, 64, true, false);
    } else {
      if (big) LAUNCH(128, 64, false, false); else LAUNCH(64, 64, false, false);
    }
  }
#undef LAUNCH
  SAMPT_CHECK_LAUNCH("gemm");
  if (p.splitk > 1) {
    long total = (long)p.M * p.N;
    hipLaunchKernelGGL(k_splitk_reduce, dim3(cdiv(total, 256)), dim3(256), 0, s, p);
    SAMPT_CHECK_LAUNCH("splitk_reduce");
  }
  return SAMPT_OK;
}

int gemm_f32(const GemmP& p, hipStream_t s) { return gemm_dispatch<float>(p, s); }
int gemm_f16(const GemmP& p, hipStream_t s) { return gemm_dispatch<half_t>(p, s); }

}  // namespace sampt
