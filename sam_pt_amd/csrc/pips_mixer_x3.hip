// The PIPS mixer's channel MLP on the fp16 matrix pipe at fp32 grade (3-term split-fp16 products, pips.py:96-128), second
// generation of pips_mixer.hip.  What round 6 measured about the exact-f32 kernel there (profiles/r6_c2_*, r6_c3_*): per block
// and workgroup 18.8 us of f32 MFMA issue (1024 x v_mfma_f32_16x16x4_f32 per wave at ~18 ns), + 11 us of stalls on its weight
// stream (each of the four waves pulls its own 256 KB through the CU's L1: 1 MB per block and CU), + 9 us fixed = 38.9 us at 32
// workgroups.  Here:
//   * products are hi.hi + hi.lo + lo.hi of fp16 pieces (v_mfma_f32_16x16x32_f16, fp32 accumulate): 96 NF MFMAs of 16 cycles per
//     wave and block instead of 1024 of 32 — weights scaled by 2^8, activations by 2^6 so that both lo pieces stay normal fp16
//     numbers, the result is multiplied by 2^-14 (exact);
//   * a workgroup owns a hidden slice for up to 64 rows: its four waves hold 16 rows each and SHARE the slice's weights, which
//     arrive once per workgroup by LDS-DMA from a host-packed LINEAR stream of 1-KB MFMA operand images (pack.pips_mixer_x3_stream:
//     every DMA instruction copies 1 KB of contiguous memory, every wave reads an image back at lane * 16 — no bank conflicts, no
//     address arithmetic, 256 / 512 KB per block and workgroup instead of 1 MB); three 32-KB stages rotate, one barrier per stage;
//   * no cross-wave reduction (the waves own different rows): a wave's 32 output fragments go straight to the slice's slab;
//   * LayerNorm, the 2^6 scaling and the hi / lo split of the MLP's input are done by the kernel that produces it
//     (k_pips_mix_pre: slab sum + residual + token mixing, ONE workgroup per sequence so that it owns whole rows), which writes
//     the operand images of the input directly; this kernel's prologue is 32 coalesced 16-byte loads per lane.
#include <type_traits>

#include "ops.h"

namespace sampt {

namespace {
constexpr int XD = 512;
// compile-time loop: the stage index must be a constant (register arrays indexed by it, a 16-iteration body hipcc will not unroll)
template <int T, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (T < N) {
    f(std::integral_constant<int, T>{});
    static_for<T + 1, N>(f);
  }
}
constexpr float X3_ASCALE = 64.0f;                       // 2^6   (pack.MIXER_X3_ASHIFT)
constexpr float X3_OSCALE = 1.0f / 16384.0f;             // 2^-14 = 2^-(8 + 6)
typedef __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
}  // namespace

// xop: operand images of the MLP input, [row fragment (16 rows)][ks 16][plane 2][lane 64][8 halves]; lane (lr, lq) of fragment
// g / 16 holds row g % 16 = lr, k = 32 ks + 8 lq + e
template <int NF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_pips_mix_mlp_x3(
    const half_t* __restrict__ xop, const half_t* __restrict__ wstream, const float* __restrict__ b1, float* __restrict__ part,
    int R, int NS) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // 4 stages x 32 KB: three in flight ahead of the multiply
  constexpr int STG = 32 * 1024, NSTG = 2 * NF;                     // NF stages of fc1 images, NF of fc2 images
  constexpr int KS_PER = 16 / NF;                                   // fc1: 32-deep k steps per stage (2 NF images each)
  constexpr int O_PER = 32 / NF;                                    // fc2: output fragments per stage (NF images each)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int slice = blockIdx.x % NS, rt = blockIdx.x / NS;
  const int r0 = rt * 64 + wave * 16;
  const bool live = r0 < R;                                         // a wave without rows still stages and keeps the barriers
  const int h0 = slice * 16 * NF;
  const char* wsrc = (const char*)(wstream + (size_t)slice * (64 * NF * 512)) + lane * 16;
  // The kernel is bound by what ONE CU can pull from memory (every weight byte is read by one workgroup once per iteration, so it
  // always comes from HBM / the Infinity Cache): 20 - 25 GB/s per CU with two stages in flight (profiles/r6_c5_*: 25.1 us for 512 KB).
  // Hence three stages in flight and non-temporal loads (aux = 2: a one-touch stream, MI355X_MICROARCH.md "nt-weights").
  auto stage = [&](int t) {                                         // images 32 t .. 32 t + 31 -> slot t % 4; this wave: wave + 4 i
    char* dst = lds + (t % 4) * STG;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int img = wave + 4 * i;
      __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + (size_t)(t * 32 + img) * 1024), (lds_void*)(dst + img * 1024), 16, 0, 2);
    }
  };
  stage(0);
  stage(1);
  stage(2);
  // ---- this wave's 16 rows as MFMA operands (already LayerNorm'ed, scaled and split by k_pips_mix_pre)
  h8 xh[16][2];
  {
    const int frag = (live ? r0 : 0) >> 4;
    const half_t* xp = xop + (size_t)frag * (16 * 2 * 512) + lane * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      xh[ks][0] = *(const h8*)(xp + (ks * 2 + 0) * 512);
      xh[ks][1] = *(const h8*)(xp + (ks * 2 + 1) * 512);
    }
  }
  float4 bq[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) bq[f] = ld4(b1 + h0 + f * 16 + lq * 4);

  f32x4 acc1[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc1[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  h8 hh[NF / 2][2];                                                 // hidden activations: [k pair][plane], see the packing's perm
  float* prow = part + ((size_t)slice * R + r0 + lr) * XD + lq * 4;
  const bool row_ok = live && r0 + lr < R;

  static_for<0, NSTG>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    // this wave's share of stage t has landed (vmcnt retires in order; younger: the eight instructions each of stages t + 1 and
    // t + 2 and this wave's output stores, which the stricter count also waits for); the barrier publishes every wave's share
    // and says that everybody is done reading stage t - 1, whose slot stage t + 3 then takes
    if (t + 2 < NSTG) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (t + 1 < NSTG) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (no LDS read in flight across the barrier: the slot of stage t - 1 is re-staged right behind it, and hipcc is free to move the
    //  wait for a stage's last operand reads below the barrier — see conv_halo_x3.hip, where exactly that was caught)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 3 < NSTG) stage(t + 3);
    const char* slot = lds + (t % 4) * STG + lane * 16;
    if (t < NF) {
      // ---- fc1: acc1[f] lane (lr, lq) reg r = 2^14 x pre-activation of hidden unit h0 + 16 f + 4 lq + r, row lr
#pragma unroll
      for (int kk = 0; kk < KS_PER; ++kk) {
        const int ks = t * KS_PER + kk;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int img = (kk * NF + f) * 2;
          const h8 whi = *(const h8*)(slot + img * 1024), wlo = *(const h8*)(slot + (img + 1) * 1024);
          acc1[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh[ks][0], acc1[f], 0, 0, 0);
          acc1[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, xh[ks][1], acc1[f], 0, 0, 0);
          acc1[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, xh[ks][0], acc1[f], 0, 0, 0);
        }
      }
      if (t == NF - 1) {
        // bias + GELU, then 2^6 and the split: k slot e of pair kp is register e of fragment 2 kp (e < 4) / e - 4 of 2 kp + 1
#pragma unroll
        for (int kp = 0; kp < NF / 2; ++kp) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int f = 2 * kp + (e >> 2), r = e & 3;
            const float b = r == 0 ? bq[f].x : (r == 1 ? bq[f].y : (r == 2 ? bq[f].z : bq[f].w));
            const float g = gelu_erf(acc1[f][r] * X3_OSCALE + b) * X3_ASCALE;
            half_t hi, lo;
            split_f16(g, hi, lo);
            hh[kp][0][e] = hi, hh[kp][1][e] = lo;
          }
        }
      }
    } else {
      // ---- fc2: two output fragments at a time (two independent accumulator chains)
#pragma unroll
      for (int oo = 0; oo < O_PER; oo += 2) {
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < NF / 2; ++kp) {
          const int i0 = (oo * (NF / 2) + kp) * 2, i1 = ((oo + 1) * (NF / 2) + kp) * 2;
          const h8 w0h = *(const h8*)(slot + i0 * 1024), w0l = *(const h8*)(slot + (i0 + 1) * 1024);
          const h8 w1h = *(const h8*)(slot + i1 * 1024), w1l = *(const h8*)(slot + (i1 + 1) * 1024);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0l, hh[kp][0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l, hh[kp][0], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h, hh[kp][1], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, hh[kp][1], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h, hh[kp][0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, hh[kp][0], a1, 0, 0, 0);
        }
        const int o = (t - NF) * O_PER + oo;
        if (row_ok) {
          *(float4*)(prow + o * 16) = make_float4(a0[0] * X3_OSCALE, a0[1] * X3_OSCALE, a0[2] * X3_OSCALE, a0[3] * X3_OSCALE);
          *(float4*)(prow + (o + 1) * 16) = make_float4(a1[0] * X3_OSCALE, a1[1] * X3_OSCALE, a1[2] * X3_OSCALE, a1[3] * X3_OSCALE);
        }
      }
    }
  });
}

// Everything between two channel MLPs for ONE sequence per workgroup (1024 threads = one float4 column of one row each):
//   x' = res + (sum of the NS slabs in order + bias)           (PART = false: x' = res — the first block of an iteration)
//   x'' = x' + token-mix(LayerNorm1(x'))                        -> xout [R][512] f32 (the next block's residual)
//   operand images of 2^6 LayerNorm2(x'') split into fp16 hi / lo  -> xop (k_pips_mix_mlp_x3's input layout)
// Token-mixing arithmetic and summation order as k_pips_token_mix / k_pips_mix_reduce<0> (four 8-unit partial sums per channel,
// (p0 + p1) + (p2 + p3)).
template <int NSB>          // slabs of the first batch (0: no slabs, x' = res; 8 or 16: all in flight together)
__global__ __launch_bounds__(1024) void k_pips_mix_pre(const float* __restrict__ part, int NS, const float* __restrict__ bias,
                                                       const float* __restrict__ res, int R, const float* __restrict__ ln1w,
                                                       const float* __restrict__ ln1b, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, const float* __restrict__ ln2w,
                                                       const float* __restrict__ ln2b, float* __restrict__ xout,
                                                       half_t* __restrict__ xop) {
  constexpr int S = 8, H = 4 * S;
  __shared__ __attribute__((aligned(16))) float xs[S][XD];
  __shared__ float psum[16], stat[S][2];
  __shared__ float sw1[H][S], sb1[H], sw2[S][H], sb2[S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq = blockIdx.x;
  const int row = tid >> 7, c4 = (tid & 127) * 4;
  const long idx = ((long)seq * S + row) * XD + c4;
  constexpr bool PART = NSB > 0;
  float4 t0[PART ? NSB : 1];
  if (PART) {
#pragma unroll
    for (int s = 0; s < NSB; ++s) t0[s] = ld4(part + (long)s * R * XD + idx);
  }
  const float4 rv = ld4(res + idx);
  const float4 bv = PART ? ld4(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 g2 = ld4(ln2w + c4), be2 = ld4(ln2b + c4);
  const int cm = tid >> 1, og = tid & 1;                        // token mixing: channel cm, hidden units 16 og .. 16 og + 15
  const float gw = ln1w[cm], gb = ln1b[cm];
  {
    const float a1 = w1[tid & 255], a2 = w2[tid & 255], a3 = b1[tid & 31], a4 = b2[tid & 7];
    if (tid < 256) sw1[tid >> 3][tid & 7] = a1, sw2[tid >> 5][tid & 31] = a2;
    if (tid < H) sb1[tid] = a3;
    if (tid < S) sb2[tid] = a4;
  }
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PART) {
#pragma unroll
    for (int s = 0; s < NSB; ++s) v.x += t0[s].x, v.y += t0[s].y, v.z += t0[s].z, v.w += t0[s].w;
    const long slab = (long)R * XD;
    for (int s0 = NSB; s0 < NS; s0 += NSB) {
      float4 t[PART ? NSB : 1];
#pragma unroll
      for (int s = 0; s < NSB; ++s) t[s] = ld4(part + (long)(s0 + s) * slab + idx);
#pragma unroll
      for (int s = 0; s < NSB; ++s) v.x += t[s].x, v.y += t[s].y, v.z += t[s].z, v.w += t[s].w;
    }
  }
  v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
  v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
  *(float4*)&xs[row][c4] = v;
  // ---- LayerNorm1 statistics (two waves per row)
  {
    const float s = wave_sum((v.x + v.y) + (v.z + v.w));
    if (lane == 0) psum[wave] = s;
  }
  __syncthreads();
  float mean = (psum[2 * row] + psum[2 * row + 1]) / (float)XD;
  __syncthreads();
  {
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float s = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    if (lane == 0) psum[wave] = s;
  }
  __syncthreads();
  if ((tid & 127) == 0) {
    stat[row][0] = mean;
    stat[row][1] = 1.0f / sqrtf((psum[2 * row] + psum[2 * row + 1]) / (float)XD + 1e-5f);
  }
  __syncthreads();
  // ---- token mixing of channel cm: two lanes, 16 hidden units each as two 8-unit partial sums
  {
    float xin[S], y[S];
#pragma unroll
    for (int t = 0; t < S; ++t) {
      xin[t] = xs[t][cm];
      y[t] = (xin[t] - stat[t][0]) * stat[t][1] * gw + gb;
    }
    float acc[S];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float pa[S];
#pragma unroll
      for (int t = 0; t < S; ++t) pa[t] = 0.f;
#pragma unroll 1
      for (int oo = 0; oo < 8; ++oo) {
        const int o = og * 16 + half * 8 + oo;
        float a = sb1[o];
#pragma unroll
        for (int t = 0; t < S; ++t) a += sw1[o][t] * y[t];
        const float h = gelu_erf(a);
#pragma unroll
        for (int t = 0; t < S; ++t) pa[t] += sw2[t][o] * h;
      }
#pragma unroll
      for (int t = 0; t < S; ++t) acc[t] = half == 0 ? pa[t] : acc[t] + pa[t];
    }
    __syncthreads();                                 // every thread has read its xin: xs can take x''
#pragma unroll
    for (int t = 0; t < S; ++t) {
      const float a = acc[t] + __shfl_xor(acc[t], 1, 64);
      if ((t & 1) == og) xs[t][cm] = xin[t] + (a + sb2[t]);
    }
  }
  __syncthreads();
  // ---- x'' out, LayerNorm2 statistics, operand images
  v = *(const float4*)&xs[row][c4];
  *(float4*)(xout + idx) = v;
  {
    const float s = wave_sum((v.x + v.y) + (v.z + v.w));
    if (lane == 0) psum[wave] = s;
  }
  __syncthreads();
  mean = (psum[2 * row] + psum[2 * row + 1]) / (float)XD;
  __syncthreads();
  float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  {
    const float s = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    if (lane == 0) psum[wave] = s;
  }
  __syncthreads();
  const float rstd = 1.0f / sqrtf((psum[2 * row] + psum[2 * row + 1]) / (float)XD + 1e-5f);
  const float y0 = (d0 * rstd * g2.x + be2.x) * X3_ASCALE, y1 = (d1 * rstd * g2.y + be2.y) * X3_ASCALE;
  const float y2 = (d2 * rstd * g2.z + be2.z) * X3_ASCALE, y3 = (d3 * rstd * g2.w + be2.w) * X3_ASCALE;
  h4 hi, lo;
  half_t a, b;
  split_f16(y0, a, b), hi[0] = a, lo[0] = b;
  split_f16(y1, a, b), hi[1] = a, lo[1] = b;
  split_f16(y2, a, b), hi[2] = a, lo[2] = b;
  split_f16(y3, a, b), hi[3] = a, lo[3] = b;
  const int grow = seq * S + row, frag = grow >> 4, olr = grow & 15;
  const int ks = c4 >> 5, olq = (c4 & 31) >> 3, e0 = c4 & 7;
  half_t* op = xop + ((size_t)(frag * 16 + ks) * 2) * 512 + (olr + 16 * olq) * 8 + e0;
  *(h4*)op = hi;
  *(h4*)(op + 512) = lo;
}

int g_pips_mixer_x3 = 1;        // sampt_pips_set_mixer(2 | 1, .): 1 = split-fp16 channel MLP (this file, the default), 0 = exact f32 (pips_mixer.hip)

size_t pips_mix_xop_halves(int nseq) { return (size_t)((nseq * 8 + 15) / 16) * 16 * 2 * 512; }

int pips_mix_mlp_x3(const half_t* xop, const half_t* wstream, const float* b1, float* part, int nseq, int NS, hipStream_t s) {
  if (nseq <= 0 || (NS != 16 && NS != 32) || !xop || !wstream || !b1 || !part) return SAMPT_ERR_ARG;
  const int R = nseq * 8, rts = (R + 63) / 64;
  constexpr int LDSB = 4 * 32 * 1024;
  dim3 grid(rts * NS), block(256);
#define MIXX(NFv)                                                                                                        \
  do {                                                                                                                   \
    static bool raised = false;                                                                                          \
    auto kern = k_pips_mix_mlp_x3<NFv>;                                                                                  \
    if (!raised) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)        \
        return SAMPT_ERR_HIP;                                                                                            \
      raised = true;                                                                                                     \
    }                                                                                                                    \
    hipLaunchKernelGGL(kern, grid, block, LDSB, s, xop, wstream, b1, part, R, NS);                                       \
  } while (0)
  if (NS == 16) MIXX(8); else MIXX(4);
#undef MIXX
  SAMPT_CHECK_LAUNCH("pips_mix_mlp_x3");
  return SAMPT_OK;
}

int pips_mix_pre(const float* part, int NS, const float* bias, const float* res, int nseq, const float* ln1w, const float* ln1b,
                 const float* tw1, const float* tb1, const float* tw2, const float* tb2, const float* ln2w, const float* ln2b,
                 float* xout, half_t* xop, hipStream_t s) {
  if (nseq <= 0 || NS < 0 || NS % 8 || (NS > 0 && (!part || !bias)) || !res || !xout || !xop || res == xout) return SAMPT_ERR_ARG;
  const int R = nseq * 8;
  dim3 grid(nseq), block(1024);
#define MIXP(NSBv) \
  hipLaunchKernelGGL(k_pips_mix_pre<NSBv>, grid, block, 0, s, part, NS, bias, res, R, ln1w, ln1b, tw1, tb1, tw2, tb2, ln2w, ln2b, xout, xop)
  if (NS == 0) MIXP(0); else if (NS % 16 == 0) MIXP(16); else MIXP(8);
#undef MIXP
  SAMPT_CHECK_LAUNCH("pips_mix_pre");
  return SAMPT_OK;
}

}  // namespace sampt
