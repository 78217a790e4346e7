// Shared definitions for the sam-pt MI355X (gfx950 / CDNA4) HIP library.
// Everything in csrc/ is written for gfx950 only: wave64, MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gelu_poly.h"

#define SAMPT_OK 0
#define SAMPT_ERR_ARG (-1)       // bad argument (shape / alignment / null pointer)
#define SAMPT_ERR_HIP (-2)       // a HIP runtime call failed (see sampt_last_error)
#define SAMPT_ERR_UNSUPPORTED (-3)
#define SAMPT_ERR_WORKSPACE (-4) // caller-provided workspace too small

typedef _Float16 half_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

namespace sampt {

void set_error(const char* where, hipError_t e);
const char* last_error();

#define SAMPT_CHECK_LAUNCH(name)                                   \
  do {                                                             \
    hipError_t _e = hipGetLastError();                             \
    if (_e != hipSuccess) { sampt::set_error(name, _e); return SAMPT_ERR_HIP; } \
  } while (0)

#define SAMPT_TRY(expr)                          \
  do {                                           \
    int _rc = (expr);                            \
    if (_rc != SAMPT_OK) return _rc;             \
  } while (0)

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_GELU_TANH = 3 };

__device__ __forceinline__ float gelu_erf(float x) {
  // exact (erf) GELU, matching torch.nn.GELU() default
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // torch.nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  (CoTracker's timm Mlp)
  const float inner = 0.79788456080286535588f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(inner));
}

__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == ACT_RELU) return x > 0.f ? x : 0.f;
  if (act == ACT_GELU) return gelu_erf(x);
  if (act == ACT_GELU_TANH) return gelu_tanh(x);
  return x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// v = hi + lo in fp16 pieces, the operand format of the 3-term split-fp16 products (conv_f16x3.hip, GemmP::x3).  Saturating:
// hi is clamped to the fp16 range and lo carries the rest, so |v| up to 2 x 65504 stays finite (beyond that lo overflows).
__device__ __forceinline__ void split_f16(float v, half_t& hi, half_t& lo) {
  hi = (half_t)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  lo = (half_t)(v - (float)hi);
}
// column c of a logical row -> its position in an x3 row (the hi half; lo is 32 halves further)
__device__ __forceinline__ long x3_col(int c) { return ((long)(c >> 5) << 6) + (c & 31); }

// XCD-aware work order of the flash attention kernels (attention.hip, attention_x3.hip).  Workgroups of a 1-D grid go round robin
// over the 8 XCDs (L & 7), each with a private 4 MiB L2.  A "unit" is the set of workgroups that read the same K / V rows —
// uh heads of one b (b = a window of a frame, or a frame) x nx query blocks — and is kept on ONE XCD: XCD x walks the units
// 8u + x, all pu = uh * nx workgroups of a unit before the next.  14 x 14 windows use uh = heads: a token's K / V of one head is
// 160 contiguous bytes inside a 7680-byte qkv row, i.e. every 128-byte line is shared by two heads and each (head, query block)
// pair used to pull it into a different L2 — 3.2 x the window's bytes for K / V alone.  Global blocks use uh = 1: the 32 query
// blocks of a (frame, head) stream the same 1.3 MB of K / V.  The units beyond the last multiple of 8 keep the plain order.
// uh == 0: plain order (x fastest, then heads, then b).
__device__ __forceinline__ void flash_wg_decode(int L, int nx, int heads, int B, int uh, int& x, int& h, int& b) {
  if (uh <= 0) {
    x = L % nx;
    const int t = L / nx;
    h = t % heads, b = t / heads;
    return;
  }
  const int hg = heads / uh, units = B * hg, pu = uh * nx, full = units & ~7;
  int unit, r;
  if (L < full * pu) {
    const int j = L >> 3;
    unit = (j / pu) * 8 + (L & 7), r = j % pu;
  } else {
    const int lt = L - full * pu;
    unit = full + lt / pu, r = lt % pu;
  }
  b = unit / hg;
  h = (unit % hg) * uh + r / nx;
  x = r % nx;
}

typedef float f32x2_g __attribute__((ext_vector_type(2)));
// ---- the two erf-GELUs of the fp16-input GEMM epilogues (gemm_f16_p8.hip, gemm_f16.hip, gemm.hip: all three kernels use the SAME
// function per output type, so a GEMM row does not depend on which kernel its launch shape selected)
// erf-GELU with the complementary error function of Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, no
// cancellation on the negative side): 0.5 x erfc(-x / sqrt 2).  The epilogue of a one-workgroup-per-CU kernel is not
// hidden behind another workgroup's MFMAs, so its VALU cost is on the critical path: 2 transcendentals + ~11 plain
// operations per element instead of the ~35 of the library erff.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = pl * t * __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);   // erfc(|z|)
  const float phi = x < 0.f ? 0.5f * e : 1.0f - 0.5f * e;
  return x * phi;
}

// erf-GELU without transcendentals for the fp16-output epilogue (gelu_poly.h, tools/gelu_poly_fit.py): x * (0.5 + xc * P(t)) with
// a degree-12 polynomial, |error| < 3.4e-6 absolute in fp32 — far below the fp16 rounding of the stored result — on PAIRS of
// elements so that the Horner chain issues as v_pk_fma_f32: ~10 VALU issue slots per element instead of ~25 for gelu_fast
// (whose two quarter-rate transcendentals cost 8 of them).  The fp32-grade modes keep gelu_fast.
__device__ __forceinline__ f32x2_g gelu_poly2(f32x2_g x) {
  constexpr float cf[GELU_POLY_DEG + 1] = GELU_POLY_COEFS;
  const f32x2_g xc = {__builtin_amdgcn_fmed3f(x[0], -GELU_POLY_C, GELU_POLY_C), __builtin_amdgcn_fmed3f(x[1], -GELU_POLY_C, GELU_POLY_C)};
  const float k = 2.0f / (GELU_POLY_C * GELU_POLY_C);
  const f32x2_g t = xc * xc * (f32x2_g){k, k} - (f32x2_g){1.f, 1.f};
  f32x2_g pl = {cf[GELU_POLY_DEG], cf[GELU_POLY_DEG]};
#pragma unroll
  for (int i = GELU_POLY_DEG - 1; i >= 0; --i) pl = pl * t + (f32x2_g){cf[i], cf[i]};
  return x * (xc * pl + (f32x2_g){0.5f, 0.5f});
}

// GELU of a half-input GEMM epilogue: fp16 result -> polynomial, f32 / x3-row result -> erfc form
__device__ __forceinline__ float gelu_half_gemm(float v, int out_f16) {
  return out_f16 == 1 ? gelu_poly2((f32x2_g){v, v})[0] : gelu_fast(v);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution (gemm.hip)
//   C[m][n] = epi( alpha * sum_k A[m][k] * W[n][k]  (+ bias[n]) ) (+ res)       (W_KN=0)
//   C[m][n] = epi( alpha * sum_k A[m][k] * W[k][n]  (+ bias[n]) ) (+ res)       (W_KN=1, f32 only)
// A may be an implicit im2col view of an NHWC tensor (conv != 0).
// ---------------------------------------------------------------------------------------------
struct GemmP {
  const void* A = nullptr;
  const void* W = nullptr;
  const void* W_lo = nullptr;    // conv_f16x3 only: the lo plane of the split weights (W = hi plane)
  const void* A_lo = nullptr;    // conv_f16x3 only: activations PRE-SPLIT into fp16 planes (A = hi plane, A_lo = lo plane, NHWC)
  const float* bias = nullptr;   // [N] or null
  const float* res = nullptr;    // residual, f32, row stride ldr, or null
  void* C = nullptr;
  const int* rowmap = nullptr;   // optional: destination row of C/res for GEMM row m (-1 = drop the row)
  const int* a_rowmap = nullptr; // optional: GEMM row m reads row a_rowmap[m] of A (gather; plain GEMMs only)
  int M = 0, N = 0, K = 0;
  int lda = 0, ldw = 0, ldc = 0, ldr = 0;
  // batching: blockIdx.z = z1 * nb2 + z2, element strides
  int nb1 = 1, nb2 = 1;
  long sA1 = 0, sA2 = 0, sW1 = 0, sW2 = 0, sC1 = 0, sC2 = 0;
  long sRowmap1 = 0;             // rowmap offset per z1
  int res_mod = 0;               // >0: residual row = row % res_mod (broadcast, e.g. positional embedding)
  // conv_f16x3 only — ConvTranspose2d(k=2, s=2) as ONE GEMM: the N = 4 * shuf_n columns are (dy, dx, channel); GEMM row
  // f*g*g + y*g + x, column block z = 2*dy + dx goes to output pixel row f*4*g*g + (2y+dy)*2g + 2x+dx (C / res have
  // shuf_n columns, bias has shuf_n entries).  shuf_g = g (0: off)
  int shuf_g = 0, shuf_n = 0;
  int act = ACT_NONE;
  int out_f16 = 0;               // half inputs only: 1 = C is half; 2 = C is half in the split "x3 row" format (see x3), ldc = 2N
  // x3 (half inputs only): fp32-grade products on the fp16 matrix pipe.  Both operands are "x3 rows": every block of 32
  // consecutive k of a row is stored as 64 halves [hi(32) | lo(32)] with v = hi + lo (hi = fp16(v), lo = fp16(v - hi);
  // weights pre-scaled by 2^F16X3_WSHIFT, the caller passes alpha = 2^-F16X3_WSHIFT).  K / lda / ldw count STORED halves
  // (2 x the real K, a multiple of 64); per 64-half K-slab the kernels issue hi.hi + hi.lo + lo.hi (the dropped lo.lo
  // term is < 2^-22 relative) instead of the two half-slab products of a plain fp16 GEMM.
  int x3 = 0;
  int w_kn = 0;                  // W stored [K][N]
  float alpha = 1.0f;
  // implicit-GEMM convolution over an NHWC input: A is [Nimg][H][W][Cin], W is [Cout][KH*KW*Cin]
  // split-K (set by the dispatcher; callers only provide the workspace): partial tiles [splitk][M][N] f32
  int xcd_swizzle = 0;           // set by the LDS-DMA fp16 launcher
  int p8_wgs = 0;                // gemm_f16_p8: persistent workgroups per XCD (0: one per CU = 32); fewer leaves whole CUs free
  int p8_stagger = 0;            // gemm_f16_p8: phase groups of the persistent workgroups (0: the process-wide setting; 1: off)
  int force_generic = 0;         // tests: bypass the specialised LDS-DMA fp16 kernel
  int splitk = 1;
  float* splitk_ws = nullptr;
  size_t splitk_ws_floats = 0;
  int conv = 0;
  int cH = 0, cW = 0, cC = 0, KH = 0, KW = 0, cstride = 1, cpad = 0, OH = 0, OW = 0;
  int cpadw = -1;                // >= 0: padding along W differs from cpad (1-D convolutions over time: KW = 1, cpadw = 0)
  // conv3x3_halo_x3 only: non-null = also write this launch's share of the following InstanceNorm's statistics — per (image, 16 x 16
  // pixel tile, channel) the pair sum(v), sum(v^2) as doubles, [nimg][conv3x3_halo_tiles(p)][N][2]: the layout k_instnorm_final reads
  double* in_part = nullptr;
  // gemm_x3_wres only — work of the NEXT kernel done on the rows while they are in registers (the mask decoder's output_upscaling):
  //   epi = 1: LayerNorm over each group of 64 output columns (= one pixel of the shuffled output; weights epi_a / epi_b [64], epi_eps),
  //            then p.act — the arithmetic of k_layernorm_rows_d64, tree for tree;
  //   epi = 2: p.act, then the dot product of each group of 32 columns with epi_a[frame * epi_ld + 0 .. 31] (frame = row / shuf_g^2):
  //            C [.] holds ONE float per output pixel (ldc = 1) — the arithmetic of k_sam_mask_dot32, tree for tree;
  //   epi = 3 (N = 256, K = 128, res): LayerNorm over the whole row after the residual (weights epi_a / epi_b [256], epi_eps) — the
  //            arithmetic of k_layernorm_rows_v4<1>, tree for tree.
  int epi = 0;
  const float* epi_a = nullptr;
  const float* epi_b = nullptr;
  float epi_eps = 0.f;
  int epi_ld = 0;
};

extern int g_p8_sched;     // gemm_f16_p8.hip: 0 = stage in the read segments (default), 1 = the round-3 schedule (sampt_gemm_set_schedule)
extern int g_p8_stagger;
extern int g_p8_trim;
extern int g_thin_min_wgs; // gemm.hip: workgroups a thin f32 GEMM must keep when it grows its tile (sampt_gemm_set_thin_min_wgs)   // gemm_f16_p8.hip: process-wide default of GemmP::p8_stagger (sampt_gemm_set_stagger)
int gemm_f32(const GemmP& p, hipStream_t s);
int gemm_f16(const GemmP& p, hipStream_t s);
// the split-K factor gemm_f32 will choose for this problem (1 = no split; needs p.splitk_ws / splitk_ws_floats set)
int gemm_f32_plan_splitk(const GemmP& p);
// fp32-grade convolution on the fp16 pipe: f32 NHWC activations, weights pre-split into fp16 hi/lo planes scaled by
// 2^F16X3_WSHIFT (conv_f16x3.hip); the caller sets alpha = 2^-F16X3_WSHIFT.  Cin % 32 == 0.
#define F16X3_WSHIFT 8
int conv_f16x3(const GemmP& p, hipStream_t s);
// conv_halo_x3.hip: the 3 x 3 stride-1 pad-1 case over pre-split planes with the input halo staged once per 16 x 16 pixel tile;
// conv_f16x3 hands eligible launches over unless g_conv_halo == 0
extern int g_conv_halo;
extern int g_halo_dbg;
extern int g_conv_in_stats;    // conv_halo_x3.hip: 1 (default) = the tracker encoder's halo convolutions also sum the following InstanceNorm's
                               // statistics (sampt_conv_set_halo(3) turns it off for A / B runs)
bool conv3x3_halo_eligible(const GemmP& p);
// gemm_x3_wres.hip: the tall short-K 1 x 1 case over f32 activations (the mask decoder's image-side projections) with the weight
// slice resident in LDS; conv_f16x3 hands eligible launches over unless g_gemm_x3_wres == 0
extern int g_gemm_x3_wres;
extern int g_gemm_x3_epi;      // 1 (default): the decoder's LayerNorm2d + GELU and mask dot product run in the weights-resident GEMMs' epilogues
                               // (sampt_gemm_set_wres(2) keeps the kernel and turns the fused tails off)
bool gemm_x3_wres_eligible(const GemmP& p);
int gemm_x3_wres(const GemmP& p, hipStream_t s);
bool gemm_x3_wres_ln_eligible(const GemmP& p);   // epi = 3: LayerNorm(res + A W^T + bias), N = 256, K = 128
int conv3x3_halo_x3(const GemmP& p, hipStream_t s);
int conv3x3_halo_tiles(const GemmP& p);

}  // namespace sampt
