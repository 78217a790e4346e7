// Tall, short-K 3-term split-fp16 GEMM with the weights RESIDENT in LDS: C [M][N] = act(alpha * A W^T + bias) + res for the mask
// decoder's image-side projections (M = frames x 4096 .. 65536 tokens, K = 64 / 128 / 256, N = 128 .. 512; engine_dec.hip
// fused_proj, L::lin, convt_pair — reference: segment_anything/modeling/transformer.py:185-232 Attention.{q,k,v,out}_proj over the
// image tokens, mask_decoder.py:53-61 output_upscaling).
//
// Why: k_conv_f16x3 (conv_f16x3.hip) gives every 128 x 128 output tile its own workgroup, which fetches the 128 rows of A (128 KB of
// f32 at K = 256) AND the 128 rows of W (128 KB of fp16 planes) — half of what a CU pulls in is the same 128 KB of weights, 6 to 12
// times per CU and launch.  These launches run at what a CU can pull through its load path with all 256 pulling (~ 20 - 25 GB/s
// each, whatever the source — L2 hits included), not at the matrix pipe's rate nor HBM's: 98 us for 200 MB, 174 us for 250 MB
// (profiles/r6_c7_clip_kernels_by_grid*).  An earlier attempt (128 rows x ALL N per workgroup, W streamed through an LDS ring)
// fetched even more W per A byte and measured equal (profiles/r6_c9_*).  Here:
//   * a workgroup owns ONE 128-column slice of W for its whole life: both planes as 1-KB MFMA operand images in LDS (K = 256:
//     128 KB of the CU's 160), loaded once by LDS-DMA;
//   * one persistent workgroup per CU, 8 waves; the slices of one row range sit on the SAME XCD at the same time (their A rows meet
//     in that XCD's L2); each wave walks its own 32-row groups with NO barrier after the weight load;
//   * A goes global -> registers (f32), is split into fp16 hi / lo there (the same saturating split as k_conv_f16x3) and is the
//     SECOND MFMA operand; 2 row fragments x 8 column fragments x 3 terms = 48 MFMAs per 32-deep K step against 16 ds_read_b128 of
//     weights (85 B / clk per CU of LDS's 128);
//   * a 4-slot register ring of "units" (32 rows x 32 k of A = 16 VGPRs, or 32 rows x 32 columns of the residual = 16 VGPRs) keeps
//     3 - 4 units per wave = 96 - 128 KB per CU in flight.  The RESIDUAL (the projected positional embedding of fused_proj, the
//     residual stream of the image -> token block) travels through the same ring as four extra units per group, and a group's
//     outputs are finished two column fragments at a time as those units arrive — an epilogue that loaded the residual where it
//     is needed would wait for everything in flight (vmcnt retires in order) once per group;
//   * the arithmetic per accumulator is the sequence of k_conv_f16x3 (per 32-deep slab: hi.lo, lo.hi, hi.hi; then alpha, bias,
//     activation, residual), so the results are BITWISE those of the kernel it replaces (tests/test_gpu_kernels.py).
#include <type_traits>

#include "ops.h"

namespace sampt {

namespace {
typedef __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(3))) void lds_void;
template <int T, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (T < N) {
    f(std::integral_constant<int, T>{});
    static_for<T + 1, N>(f);
  }
}

// hipcc moves the (side-effect-free) split of loop-carried ring registers to the top of the loop body — and with it the wait for
// every unit in flight.  An empty volatile asm that "modifies" a unit's registers keeps its consumption where it is written.
__device__ __forceinline__ void pin(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// KS = K / 32.  RES: p.res travels through the ring (no pixel shuffle then).  Grid: 256 workgroups; workgroup b sits on XCD b & 7,
// q = b >> 3 is its number there: column slice q % S of row lane q / S (RL = 32 / S row lanes per XCD; workgroups beyond S * RL idle).
// ACT: ACT_NONE or ACT_GELU (a run-time switch inside the unrolled epilogue would be 64 copies of every activation).
// EPI: GemmP::epi (0: plain; 1: LayerNorm over 64-column groups, then ACT; 2: ACT, then the 32-column dot product with a per-frame vector)
template <int KS, bool RES, int ACT, int EPI>
__global__ __launch_bounds__(512) void k_gemm_x3_wres(GemmP p, int S, int RL) {
  constexpr int K = KS * 32;
  constexpr int U = KS + (RES ? 4 : 0);        // units of a 32-row group
  constexpr int GB = (U % 4) ? 2 : 1;          // groups per loop body: the ring slot of a unit (its number % 4) must be a constant
  constexpr int UB = U * GB;
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // [W images (ks, j, plane): KS * 16 KB | bias 512 B | epilogue table]
  char* const w_lds = lds;
  float* const bias_lds = (float*)(lds + KS * 16 * 1024);
  float* const epi_lds = bias_lds + 128;       // EPI 1: gamma [64] | beta [64]; EPI 2: the per-frame vectors [frames][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int sl = q % S, rl = q / S;
  if (rl >= RL) return;
  const int n0 = sl * 128;
  const int M = p.M, N = p.N;
  const int WL = (rl * 8 + xcd) * 8 + wave, NWL = RL * 64;           // this wave's lane among all waves of the slice
  const int G = (M + 31) >> 5;
  const int ng = WL < G ? (G - WL + NWL - 1) / NWL : 0;              // its groups: WL, WL + NWL, ...
  const float* __restrict__ A = (const float*)p.A;
  const float* __restrict__ res = p.res;

  float4 ring[4][4];
  int rrow[2] = {0, 0};
  // unit (group gidx, position pos): pos < KS: A[32 rows][32 pos .. + 32], lane (lr, lq) of fragment i holds k = 8 lq .. + 8;
  // pos >= KS: res[32 rows][n0 + 32 (pos - KS) .. + 32], lane holds the 4 columns 4 lq .. + 4 of fragment j = 2 (pos - KS) + jj
  auto issue = [&](auto slotc, auto posc, int gidx) {
    constexpr int slot = decltype(slotc)::value, pos = decltype(posc)::value;
    const int g = WL + NWL * (gidx < ng ? gidx : ng - 1);            // past the end: the last group again (loaded, never used)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row = g * 32 + 16 * i + lr;
      row = row < M ? row : M - 1;
      if constexpr (pos < KS) {
        const float* pa = A + (long)row * K + pos * 32 + 8 * lq;
        ring[slot][2 * i] = *(const float4*)pa;
        ring[slot][2 * i + 1] = *(const float4*)(pa + 4);
      } else {
        constexpr int rr = pos - KS;
        if (rr == 0) rrow[i] = p.res_mod > 0 ? row % p.res_mod : row;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int col = n0 + 16 * (2 * rr + jj) + 4 * lq;
          ring[slot][2 * i + jj] = *(const float4*)(res + (long)rrow[i] * p.ldr + (col < N ? col : 0));
        }
      }
    }
  };
  if (ng > 0) static_for<0, 4>([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    issue(uc, std::integral_constant<int, u % U>{}, u / U);
  });

  // ---- the weight slice, once: image (ks, j, plane) = W_plane[n0 + 16 j + lr][32 ks + 8 lq .. + 8], lane-linear (what LDS-DMA writes)
  {
    const char* Wh = (const char*)p.W;
    const char* Wl = (const char*)p.W_lo;
    for (int im = wave; im < KS * 16; im += 8) {
      const int pl = im & 1, j = (im >> 1) & 7, ks = im >> 4;
      int n = n0 + 16 * j + lr;
      n = n < N ? n : N - 1;
      const char* src = (pl ? Wl : Wh) + ((long)n * p.ldw + 32 * ks + 8 * lq) * 2;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(w_lds + im * 1024), 16, 0, 0);
    }
    if (tid < 32) {
      const int col = n0 + 4 * tid;                                  // (pixel shuffle: the bias has shuf_n entries, column % shuf_n)
      ((float4*)bias_lds)[tid] = (p.bias && col < N) ? *(const float4*)(p.bias + (p.shuf_g ? col % p.shuf_n : col))
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (EPI == 1 && tid < 128) epi_lds[tid] = tid < 64 ? p.epi_a[tid] : p.epi_b[tid - 64];
    if (EPI == 2) {
      const int nfr = M / (p.shuf_g * p.shuf_g);
      for (int i = tid; i < nfr * 32; i += 512) epi_lds[i] = p.epi_a[(long)(i >> 5) * p.epi_ld + (i & 31)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (ng == 0) return;

  // output addressing: element offset of (fragment j, this lane's 4 columns) relative to its row's first element
  int joff[8];                                  // (uniform: kept in scalar registers)
  const int sg = p.shuf_g, sP = sg * sg;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c0 = n0 + 16 * j;
    if (sg) {
      const int z = c0 / p.shuf_n;
      joff[j] = __builtin_amdgcn_readfirstlane(((z >> 1) * 2 * sg + (z & 1)) * p.ldc + (c0 - z * p.shuf_n));
    } else {
      joff[j] = c0;
    }
  }
  auto out_row = [&](int row) -> long {         // destination pixel row (the pixel shuffle of a transposed convolution, see GemmP)
    if (!sg) return row;
    const int f = row / sP, rem = row - f * sP, y = rem / sg, x = rem - y * sg;
    return (long)f * 4 * sP + (long)(2 * y) * 2 * sg + 2 * x;
  };
  const float alpha = p.alpha;
  float* __restrict__ C = (float*)p.C;

  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto finish = [&](int i, int j, long orow, bool ok, const float4* r) {
    const float4 b = ((const float4*)bias_lds)[4 * j + lq];
    float4 v = make_float4(acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha);
    v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
    v.x = apply_act(v.x, ACT), v.y = apply_act(v.y, ACT), v.z = apply_act(v.z, ACT), v.w = apply_act(v.w, ACT);
    if (r) v.x += r->x, v.y += r->y, v.z += r->z, v.w += r->w;
    if (ok && n0 + 16 * j + 4 * lq < N) *(float4*)(C + orow * p.ldc + joff[j] + 4 * lq) = v;
    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  for (int gb0 = 0; gb0 < ng; gb0 += GB) {
    static_for<0, UB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int gi = u / U, pos = u % U, slot = u % 4;
      const int gidx = gb0 + gi;
      std::integral_constant<int, slot> slotc;
      std::integral_constant<int, (u + 4) % U> npos;
#pragma unroll
      for (int e = 0; e < 4; ++e) pin(ring[slot][e]);
      if constexpr (pos < KS) {
        h8 ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float4 v0 = ring[slot][2 * i], v1 = ring[slot][2 * i + 1];
          const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            half_t a, b;
            split_f16(vv[e], a, b);
            ah[i][e] = a, al[i][e] = b;
          }
        }
        issue(slotc, npos, gb0 + (u + 4) / U);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const h8 bh = *(const h8*)(w_lds + ((pos * 8 + j) * 2) * 1024 + lane * 16);
          const h8 bl = *(const h8*)(w_lds + ((pos * 8 + j) * 2 + 1) * 1024 + lane * 16);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], acc[i][j], 0, 0, 0);
        }
        if constexpr (!RES && pos == KS - 1) {                          // no residual: the whole group is finished here
          const int g = WL + NWL * gidx;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = g * 32 + 16 * i + lr;
            const bool ok = gidx < ng && row < M;
            const long orow = out_row(ok ? row : 0);
            if constexpr (EPI == 0) {
#pragma unroll
              for (int j = 0; j < 8; ++j) finish(i, j, orow, ok, nullptr);
            } else if constexpr (EPI == 1) {
              // LayerNorm over the 64 columns of a shuffled pixel = 4 fragments x the 4 lq lanes of this row: lane (lr, lq) of
              // fragment jj stands where thread 4 jj + lq of k_layernorm_rows_d64 stands, and the sums follow row16_sum's tree
              // (xor 1, xor 2 across lq = lanes ^ 16, ^ 32; then quads 0 + 1, 2 + 3; then halves), so the result is that kernel's
#pragma unroll
              for (int grp = 0; grp < 2; ++grp) {
                float4 v[4];
                float q[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                  const int j = 4 * grp + jj;
                  const float4 b = ((const float4*)bias_lds)[4 * j + lq];
                  v[jj] = make_float4(acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha);
                  v[jj].x += b.x, v[jj].y += b.y, v[jj].z += b.z, v[jj].w += b.w;          // (the two steps of finish())
                  acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                  float x = (v[jj].x + v[jj].y) + (v[jj].z + v[jj].w);
                  x += __shfl_xor(x, 16, 64);
                  x += __shfl_xor(x, 32, 64);
                  q[jj] = x;
                }
                const float mean = ((q[0] + q[1]) + (q[2] + q[3])) * (1.0f / 64.0f);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                  v[jj].x -= mean, v[jj].y -= mean, v[jj].z -= mean, v[jj].w -= mean;
                  float x = __builtin_fmaf(v[jj].y, v[jj].y, v[jj].x * v[jj].x) + __builtin_fmaf(v[jj].w, v[jj].w, v[jj].z * v[jj].z);
                  x += __shfl_xor(x, 16, 64);
                  x += __shfl_xor(x, 32, 64);
                  q[jj] = x;
                }
                const float rstd = 1.0f / sqrtf(((q[0] + q[1]) + (q[2] + q[3])) * (1.0f / 64.0f) + p.epi_eps);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                  const int j = 4 * grp + jj;
                  const float4 w4 = ((const float4*)epi_lds)[4 * jj + lq], b4 = ((const float4*)epi_lds)[16 + 4 * jj + lq];
                  const float4 o = make_float4(apply_act(__builtin_fmaf(v[jj].x * rstd, w4.x, b4.x), ACT),
                                               apply_act(__builtin_fmaf(v[jj].y * rstd, w4.y, b4.y), ACT),
                                               apply_act(__builtin_fmaf(v[jj].z * rstd, w4.z, b4.z), ACT),
                                               apply_act(__builtin_fmaf(v[jj].w * rstd, w4.w, b4.w), ACT));
                  if (ok && n0 + 16 * j + 4 * lq < N) *(float4*)(C + orow * p.ldc + joff[j] + 4 * lq) = o;
                }
              }
            } else {
              // ACT, then <row's 32 columns of a shuffled pixel, the frame's vector>: 2 fragments x the 4 lq lanes stand where the 8
              // threads of a pixel stand in k_sam_mask_dot32, and the sum follows oct_sum's tree (xor 1, xor 2, then quads 0 + 1)
              const int fr = (ok ? row : 0) / sP;
#pragma unroll
              for (int zg = 0; zg < 4; ++zg) {
                float q[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                  const int j = 2 * zg + jj;
                  const float4 b = ((const float4*)bias_lds)[4 * j + lq];
                  float4 t = make_float4(acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha);
                  t.x += b.x, t.y += b.y, t.z += b.z, t.w += b.w;
                  t.x = apply_act(t.x, ACT), t.y = apply_act(t.y, ACT), t.z = apply_act(t.z, ACT), t.w = apply_act(t.w, ACT);
                  acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                  const float4 h = ((const float4*)epi_lds)[fr * 8 + 4 * jj + lq];
                  float x = __builtin_fmaf(h.w, t.w, __builtin_fmaf(h.z, t.z, __builtin_fmaf(h.y, t.y, h.x * t.x)));
                  x += __shfl_xor(x, 16, 64);
                  x += __shfl_xor(x, 32, 64);
                  q[jj] = x;
                }
                if (ok && lq == 0 && n0 + 32 * zg < N) C[orow * p.ldc + joff[2 * zg]] = q[0] + q[1];
              }
            }
          }
        }
      } else {                                                          // a residual unit: two column fragments of both row fragments
        constexpr int rr = pos - KS;
        const int g = WL + NWL * gidx;
        float4 r4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r4[e] = ring[slot][e];
        issue(slotc, npos, gb0 + (u + 4) / U);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = g * 32 + 16 * i + lr;
          const bool ok = gidx < ng && row < M;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) finish(i, 2 * rr + jj, ok ? row : 0, ok, &r4[2 * i + jj]);
        }
      }
    });
  }
}

// ---------------------------------------------------------------------------------------------
// out = LayerNorm(res + A W^T + bias) for N = 256, K = 128: the tail of the decoder's image -> token attention block
// (transformer.py:145-150: keys = norm4(keys + attn_out)).  All 256 columns of a row sit in ONE workgroup (W: 256 x 128 x 2 planes
// = 128 KB of LDS), so the LayerNorm that used to be a pass of its own over the 100 MB the GEMM had just written runs on the
// registers: a wave owns 16-row groups (64 accumulator registers), the A and residual units (16 rows x 32 k / x 32 columns, 8
// registers) go round a 6-slot register ring, the residual units finish two column fragments each into the accumulator registers
// and the last one normalises the row.  The sums follow k_layernorm_rows_v4<1>'s wave_sum tree — thread 4 j + lq of that kernel is
// lane (lr, lq) of fragment j here: xor 32 / 16 / 8 / 4 pair the fragments j ^ 8, ^ 4, ^ 2, ^ 1 in registers, xor 2 / 1 the lanes
// ^ 32, ^ 16 — and its arithmetic operation for operation, so a row is the same bits whichever kernel normalised it.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pin2(float4& a, float4& b) {
  asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
}

__global__ __launch_bounds__(512) void k_gemm_x3_wres_ln(GemmP p) {
  constexpr int KS = 4, K = 128, NJ = 16, U = KS + 8, NS = 6;   // 12 units per 16-row group, 6 ring slots (slot = unit % 6)
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // [W images (ks, j, plane): 128 KB | bias 1 KB | gamma 1 KB | beta 1 KB]
  char* const w_lds = lds;
  float* const bias_lds = (float*)(lds + KS * NJ * 2 * 1024);
  float* const gam_lds = bias_lds + 256;
  float* const bet_lds = gam_lds + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int M = p.M;
  const int WL = blockIdx.x * 8 + wave, NWL = gridDim.x * 8;
  const int G = (M + 15) >> 4;
  const int ng = WL < G ? (G - WL + NWL - 1) / NWL : 0;
  const float* __restrict__ A = (const float*)p.A;
  const float* __restrict__ res = p.res;

  float4 ring[NS][2];
  auto issue = [&](auto slotc, auto posc, int gidx) {
    constexpr int slot = decltype(slotc)::value, pos = decltype(posc)::value;
    const int g = WL + NWL * (gidx < ng ? gidx : ng - 1);
    int row = g * 16 + lr;
    row = row < M ? row : M - 1;
    if constexpr (pos < KS) {
      const float* pa = A + (long)row * K + pos * 32 + 8 * lq;
      ring[slot][0] = *(const float4*)pa;
      ring[slot][1] = *(const float4*)(pa + 4);
    } else {
      constexpr int rr = pos - KS;
      const float* pr = res + (long)row * p.ldr + 32 * rr + 4 * lq;
      ring[slot][0] = *(const float4*)pr;
      ring[slot][1] = *(const float4*)(pr + 16);
    }
  };
  if (ng > 0) static_for<0, NS>([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    issue(uc, std::integral_constant<int, u % U>{}, u / U);
  });
  {
    const char* Wh = (const char*)p.W;
    const char* Wl = (const char*)p.W_lo;
    for (int im = wave; im < KS * NJ * 2; im += 8) {
      const int pl = im & 1, j = (im >> 1) & 15, ks = im >> 5;
      const char* src = (pl ? Wl : Wh) + ((long)(16 * j + lr) * p.ldw + 32 * ks + 8 * lq) * 2;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(w_lds + im * 1024), 16, 0, 0);
    }
    if (tid < 256) bias_lds[tid] = p.bias ? p.bias[tid] : 0.f, gam_lds[tid] = p.epi_a[tid], bet_lds[tid] = p.epi_b[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (ng == 0) return;
  const float alpha = p.alpha, eps = p.epi_eps;
  float* __restrict__ C = (float*)p.C;

  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // sum over the row of one value per (fragment, lane): wave_sum's pairing order (see the header)
  auto row_sum = [&](float (&x)[NJ]) -> float {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += x[j + 8];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] += x[j + 4];
    x[0] += x[2], x[1] += x[3];
    float t = x[0] + x[1];
    t += __shfl_xor(t, 32, 64);
    t += __shfl_xor(t, 16, 64);
    return t;
  };

  for (int gb0 = 0; gb0 < ng; ++gb0) {
    static_for<0, U>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int gi = 0, pos = u, slot = u % NS;
      const int gidx = gb0 + gi;
      std::integral_constant<int, slot> slotc;
      std::integral_constant<int, (u + NS) % U> npos;
      pin2(ring[slot][0], ring[slot][1]);
      if constexpr (pos < KS) {
        h8 ah, al;
        const float4 v0 = ring[slot][0], v1 = ring[slot][1];
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          half_t a, b;
          split_f16(vv[e], a, b);
          ah[e] = a, al[e] = b;
        }
        issue(slotc, npos, gb0 + (u + NS) / U);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const h8 bh = *(const h8*)(w_lds + ((pos * NJ + j) * 2) * 1024 + lane * 16);
          const h8 bl = *(const h8*)(w_lds + ((pos * NJ + j) * 2 + 1) * 1024 + lane * 16);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[j], 0, 0, 0);
        }
      } else {
        constexpr int rr = pos - KS;
        const float4 r4[2] = {ring[slot][0], ring[slot][1]};
        issue(slotc, npos, gb0 + (u + NS) / U);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {                                // finish(): alpha, bias, (no activation), residual
          const int j = 2 * rr + jj;
          const float4 b = ((const float4*)bias_lds)[4 * j + lq];
          float4 v = make_float4(acc[j][0] * alpha, acc[j][1] * alpha, acc[j][2] * alpha, acc[j][3] * alpha);
          v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
          v.x += r4[jj].x, v.y += r4[jj].y, v.z += r4[jj].z, v.w += r4[jj].w;
          acc[j] = (f32x4){v.x, v.y, v.z, v.w};
        }
        if constexpr (rr == 7) {                                        // the row is complete: LayerNorm, store
          const int row = (WL + NWL * gidx) * 16 + lr;
          const bool ok = gidx < ng && row < M;
          float x[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) x[j] = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
          const float mean = row_sum(x) / 256.0f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const float d0 = acc[j][0] - mean, d1 = acc[j][1] - mean, d2 = acc[j][2] - mean, d3 = acc[j][3] - mean;
            x[j] = __builtin_fmaf(d1, d1, d0 * d0) + __builtin_fmaf(d3, d3, d2 * d2);
          }
          const float rstd = 1.0f / sqrtf(row_sum(x) / 256.0f + eps);
          float* orow = C + (long)(ok ? row : 0) * p.ldc + 4 * lq;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const float4 w4 = ((const float4*)gam_lds)[4 * j + lq], b4 = ((const float4*)bet_lds)[4 * j + lq];
            const float4 o = make_float4(__builtin_fmaf((acc[j][0] - mean) * rstd, w4.x, b4.x), __builtin_fmaf((acc[j][1] - mean) * rstd, w4.y, b4.y),
                                         __builtin_fmaf((acc[j][2] - mean) * rstd, w4.z, b4.z), __builtin_fmaf((acc[j][3] - mean) * rstd, w4.w, b4.w));
            if (ok) *(float4*)(orow + 16 * j) = o;
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    });
  }
}
}  // namespace

int g_gemm_x3_epi = 1;
int g_gemm_x3_wres = 1;     // sampt_gemm_set_wres: 0 = these launches stay on k_conv_f16x3 (A / B)

// f32 A with contiguous K = 64 / 128 / 256 rows (a 1 x 1 "convolution"), enough rows to fill the persistent grid
bool gemm_x3_wres_eligible(const GemmP& p) {
  if (!p.conv || p.A_lo || p.KH != 1 || p.KW != 1 || p.cstride != 1 || p.cpad != 0) return false;
  if (p.K != p.cC || (p.K != 64 && p.K != 128 && p.K != 256)) return false;
  if (p.M < 16384 || p.N < 64 || p.N > 1024 || p.N % 4 || (p.epi != 2 && p.ldc % 4)) return false;
  if (p.shuf_g && (p.res || p.shuf_n % 16)) return false;
  if (p.act != ACT_NONE && (p.act != ACT_GELU || p.res)) return false;
  if (p.res && p.ldr % 4) return false;
  if (p.epi == 3) return gemm_x3_wres_ln_eligible(p);
  if (p.epi) {   // the fused tails of the decoder's output_upscaling, at the shapes they have there
    if (p.res || !p.shuf_g || !p.epi_a || p.act != ACT_GELU) return false;
    if (p.epi == 1 && (p.K != 256 || p.shuf_n != 64 || !p.epi_b)) return false;
    if (p.epi == 2 && (p.K != 64 || p.shuf_n != 32 || p.ldc != 1 || p.M / (p.shuf_g * p.shuf_g) > 128 || p.epi_ld < 32)) return false;
    if (p.epi != 1 && p.epi != 2) return false;
  }
  return true;
}

// epi = 3: out = LayerNorm(res + A W^T + bias) over the whole 256-column row (k_gemm_x3_wres_ln)
bool gemm_x3_wres_ln_eligible(const GemmP& p) {
  return p.epi == 3 && p.conv && !p.A_lo && p.KH == 1 && p.KW == 1 && p.cstride == 1 && p.cpad == 0 && p.K == 128 && p.cC == 128 && p.N == 256 &&
         p.M >= 16384 && p.res && !p.res_mod && !p.shuf_g && p.act == ACT_NONE && p.ldc % 4 == 0 && p.ldr % 4 == 0 && p.ldw == 128 && p.epi_a &&
         p.epi_b && !(((uintptr_t)p.A | (uintptr_t)p.res | (uintptr_t)p.C | (uintptr_t)p.W | (uintptr_t)p.W_lo) & 15);
}

int gemm_x3_wres_ln(const GemmP& p, hipStream_t s) {
  constexpr int LDSB = 4 * 16 * 2 * 1024 + 3 * 1024;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute((const void*)k_gemm_x3_wres_ln, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess) return SAMPT_ERR_HIP;
    raised = true;
  }
  hipLaunchKernelGGL(k_gemm_x3_wres_ln, dim3(256), dim3(512), LDSB, s, p);
  SAMPT_CHECK_LAUNCH("gemm_x3_wres_ln");
  return SAMPT_OK;
}

int gemm_x3_wres(const GemmP& p, hipStream_t s) {
  if (p.epi == 3) return gemm_x3_wres_ln_eligible(p) ? gemm_x3_wres_ln(p, s) : SAMPT_ERR_UNSUPPORTED;
  const int S = cdiv(p.N, 128), RL = 32 / S;
  const int KS = p.K / 32;
  const int epib = p.epi == 1 ? 512 : (p.epi == 2 ? (p.M / (p.shuf_g * p.shuf_g)) * 128 : 0);
  const int ldsb = KS * 16 * 1024 + 512 + epib;
#define WRES(KSv, RESv, ACTv, EPIv)                                                                                      \
  do {                                                                                                                   \
    static bool raised = false;                                                                                          \
    auto kern = k_gemm_x3_wres<KSv, RESv, ACTv, EPIv>;                                                                   \
    if (!raised) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,                             \
                              KSv * 16 * 1024 + 512 + (EPIv == 2 ? 128 * 128 : 512)) != hipSuccess)                      \
        return SAMPT_ERR_HIP;                                                                                            \
      raised = true;                                                                                                     \
    }                                                                                                                    \
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), ldsb, s, p, S, RL);                                                   \
  } while (0)
  if (p.epi == 1) WRES(8, false, ACT_GELU, 1);
  else if (p.epi == 2) WRES(2, false, ACT_GELU, 2);
  else if (p.res) {
    if (KS == 2) WRES(2, true, ACT_NONE, 0);
    else if (KS == 4) WRES(4, true, ACT_NONE, 0);
    else WRES(8, true, ACT_NONE, 0);
  } else if (p.act == ACT_GELU) {
    if (KS == 2) WRES(2, false, ACT_GELU, 0);
    else if (KS == 4) WRES(4, false, ACT_GELU, 0);
    else WRES(8, false, ACT_GELU, 0);
  } else {
    if (KS == 2) WRES(2, false, ACT_NONE, 0);
    else if (KS == 4) WRES(4, false, ACT_NONE, 0);
    else WRES(8, false, ACT_NONE, 0);
  }
#undef WRES
  SAMPT_CHECK_LAUNCH("gemm_x3_wres");
  return SAMPT_OK;
}

}  // namespace sampt
