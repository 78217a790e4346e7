// The MLP-Mixer of PIPS's DeltaBlock (pips.py:96-128, 290-317) as TWO launches per mixer block instead of four, sized for a
// handful of CUs: the window rounds run beside the ViT encoder (sam_pt.py SamPt.forward), whose persistent GEMM workgroups own
// every CU they get, so what the chain costs the clip is the CUs it holds and how long it holds them.
//
// Rows of the mixer are (point, frame): R = n * 8, D = 512 channels, 2048 hidden units, fp32 weights (8 MB per block).
//
//   k_pips_mix_mlp<NF>      channel-mixing half of block i WITHOUT its final sum:  part[slice] = fc2_slice(gelu(fc1_slice(LN2(x''))))
//       grid = row groups (16 rows = 2 sequences) x NS hidden slices, 256 threads.  The four waves of a workgroup own the four
//       quarters of the slice's hidden units and ALL 16 rows.  A wave keeps its LayerNorm'ed 16 x 512 rows in registers as the
//       MFMA operand (v_mfma_f32_16x16x4_f32, exact fp32), streams its W1 rows straight from L2 into the other operand
//       (software-pipelined 16-byte loads, 8 chunks in flight), and the C^T fragments it gets — lane (r, q) holds hidden units
//       4q .. 4q+3 of row r — ARE the operand layout of the second product, so the hidden activations never leave registers:
//       bias + GELU in place, then 32 output fragments against its W2 columns.  The waves' partial sums meet in LDS (fixed
//       order), one slab [R][512] per slice goes to memory.
//   k_pips_mix_reduce<MODE> everything between two channel MLPs, grid = (sequence, 64-channel chunk):
//       x' = x''_prev + (sum over slices in order + b2)  for the sequence's 8 full rows (the LayerNorm statistics need them),
//       MODE 0: token-mixing block i + 1 on the chunk's channels (pips.py:116, 120-121; arithmetic and summation order of
//               k_pips_token_mix) -> x'' ; MODE 1: final LayerNorm + mean over the 8 tokens (pips.py:125-126).
//
// Why launches and not one persistent kernel with in-kernel hand-offs: every seam here is an all-to-all over the 64 x 512
// activations; on this part a flag hand-off between loaded CUs costs 3 - 5 us against 1.5 - 1.9 us for a kernel boundary
// (MI355X_MICROARCH.md price list: handoff-flag, boundary; cdna_hip_programming.md 5.6: "cut at every all-to-all seam").
// Why 2-D (row group x hidden slice) and not all rows per workgroup: the slab a slice writes is 16 rows instead of 64, so the
// reduce kernel reads NS x 16 KB per sequence instead of 4 NS x 16 KB; the price is that each of the 4 row groups streams the
// weights (1 MB per workgroup and block from L2 — same-slice workgroups are dealt to the same XCD: blockIdx % 8 == slice % 8),
// hidden behind 13.6 us of f32 MFMA work per workgroup and block at NS = 8.
#include "ops.h"

namespace sampt {

namespace {
constexpr int MD = 512, MH = 2048;

__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
}  // namespace

// hipcc sinks every weight load to just before its first use (one exposed L2 round trip per chunk) unless the program order
// is pinned: a scheduling barrier after each stage's load issue and after its MFMAs; one wave per SIMD is all this kernel
// wants (declared, so that keeping 8 chunks of weights in flight is not "register pressure" to the scheduler).
#define MIX_PIN() __builtin_amdgcn_sched_barrier(0)

template <int NF, int DIAG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_pips_mix_mlp(const float* __restrict__ x, const float* __restrict__ lnw,
                                                      const float* __restrict__ lnb, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      float* __restrict__ part, int R, int NS) {
  // fc1 runs in steps of two 16-deep K chunks = ONE 128-byte line of every weight row: the two loads that split a line are
  // issued back to back (the second hits the line the first requested).  With one chunk per step they were a whole step's MFMAs
  // (and the other waves' 256 lines: the whole 32 KiB L1) apart, so every line crossed L2 -> L1 twice: 38.9 us per launch at 32
  // workgroups against 13.7 us of MFMA issue (profiles/r6_c2_*).
  constexpr int PD = 4;                  // fc1: steps whose weight loads are in flight (8 chunks)
  constexpr int OG = 4;                  // fc2: output fragments per group (one per wave in the LDS reduction)
  __shared__ f32x4 red[2][4][OG][64];    // 32 KB, double-buffered: one barrier per group
  __shared__ __attribute__((aligned(16))) float lng[2][MD];   // LayerNorm weight / bias: read per chunk as LDS broadcasts
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int slice = blockIdx.x % NS, rg = blockIdx.x / NS;
  const int r0 = rg * 16;
  const int row = r0 + lr < R ? r0 + lr : R - 1;
  const int h0 = (slice * 4 + wave) * (16 * NF);
  const float* w1p = w1 + (long)(h0 + lr) * MD + lq * 4;      // + f * 16 * MD + c * 16
  const float* w2p = w2 + (long)lr * MH + h0 + lq * 4;        // + o * 16 * MH + f * 16

  const float4 lnv = ld4((tid < 128 ? lnw : lnb) + (tid & 127) * 4);     // first in the queue: vmcnt retires in order
  float4 wq[PD][NF][2];
#pragma unroll
  for (int p = 0; p < PD; ++p)
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      wq[p][f][0] = ld4(w1p + (long)f * 16 * MD + p * 32);
      wq[p][f][1] = ld4(w1p + (long)f * 16 * MD + p * 32 + 16);
    }
  float4 bq[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) bq[f] = ld4(b1 + h0 + f * 16 + lq * 4);

  // ---- LayerNorm of the wave's 16 rows, in the operand layout: lane (lr, lq) holds k = 16 c + 4 lq + j of row lr
  float4 xa[32];
  {
    const float* xr = x + (long)row * MD + lq * 4;
#pragma unroll
    for (int c = 0; c < 32; ++c) xa[c] = ld4(xr + c * 16);
    MIX_PIN();
    *(float4*)&lng[tid >> 7][(tid & 127) * 4] = lnv;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) sum += (xa[c].x + xa[c].y) + (xa[c].z + xa[c].w);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)MD;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float d0 = xa[c].x - mean, d1 = xa[c].y - mean, d2 = xa[c].z - mean, d3 = xa[c].w - mean;
      sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)MD + 1e-5f);
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 32; cb += 8) {     // 16 LDS reads in flight per batch (left alone hipcc waits for every pair)
      float4 g[8], b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        g[i] = *(const float4*)&lng[0][(cb + i) * 16 + lq * 4];
        b[i] = *(const float4*)&lng[1][(cb + i) * 16 + lq * 4];
      }
      MIX_PIN();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = cb + i;
        xa[c].x = (xa[c].x - mean) * rstd * g[i].x + b[i].x;
        xa[c].y = (xa[c].y - mean) * rstd * g[i].y + b[i].y;
        xa[c].z = (xa[c].z - mean) * rstd * g[i].z + b[i].z;
        xa[c].w = (xa[c].w - mean) * rstd * g[i].w + b[i].w;
      }
      MIX_PIN();
    }
  }
  MIX_PIN();

  // ---- fc1: acc1[f] lane (lr, lq) reg r = pre-activation of hidden unit h0 + 16 f + 4 lq + r for row lr
  float4 w2q[3][OG][NF];                 // fc2 weights, three groups rotating (two in flight ahead of the multiply)
  auto load_w2 = [&](int g, int slot) {
#pragma unroll
    for (int i = 0; i < OG; ++i)
#pragma unroll
      for (int f = 0; f < NF; ++f) w2q[slot][i][f] = ld4(w2p + (long)(g * OG + i) * 16 * MH + f * 16);
  };
  f32x4 acc1[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc1[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c2 = 0; c2 < 16; ++c2) {
    float4 wv[NF][2];
#pragma unroll
    for (int f = 0; f < NF; ++f) wv[f][0] = wq[c2 % PD][f][0], wv[f][1] = wq[c2 % PD][f][1];
    if (c2 + PD < 16 && !DIAG) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        wq[c2 % PD][f][0] = ld4(w1p + (long)f * 16 * MD + (c2 + PD) * 32);
        wq[c2 % PD][f][1] = ld4(w1p + (long)f * 16 * MD + (c2 + PD) * 32 + 16);
      }
    }
    if (c2 == 16 - PD) load_w2(0, 0);    // the W1 stream has ended: start the W2 stream under the last steps
    if (c2 == 16 - PD / 2) load_w2(1, 1);
    MIX_PIN();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 2 * c2 + e;
#pragma unroll
      for (int f = 0; f < NF; ++f) acc1[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[f][e].x, xa[c].x, acc1[f], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NF; ++f) acc1[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[f][e].y, xa[c].y, acc1[f], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NF; ++f) acc1[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[f][e].z, xa[c].z, acc1[f], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NF; ++f) acc1[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[f][e].w, xa[c].w, acc1[f], 0, 0, 0);
    }
    MIX_PIN();
  }
  float gh[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const float4 b = bq[f];
    gh[f][0] = gelu_erf(acc1[f][0] + b.x);
    gh[f][1] = gelu_erf(acc1[f][1] + b.y);
    gh[f][2] = gelu_erf(acc1[f][2] + b.z);
    gh[f][3] = gelu_erf(acc1[f][3] + b.w);
  }

  // ---- fc2 over this wave's hidden units: acc2[i] lane (lr, lq) reg r = partial of output column 16 o + 4 lq + r, row lr
  constexpr int NG = MD / 16 / OG;       // 8 groups
  float* prow = part + ((long)slice * R + r0 + lr) * MD + lq * 4;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 2 < NG && !DIAG) load_w2(g + 2, (g + 2) % 3);
    MIX_PIN();
    f32x4 acc2[OG];
#pragma unroll
    for (int i = 0; i < OG; ++i) acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int i = 0; i < OG; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2q[g % 3][i][f].x, gh[f][0], acc2[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < OG; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2q[g % 3][i][f].y, gh[f][1], acc2[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < OG; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2q[g % 3][i][f].z, gh[f][2], acc2[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < OG; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2q[g % 3][i][f].w, gh[f][3], acc2[i], 0, 0, 0);
    }
    MIX_PIN();
#pragma unroll
    for (int i = 0; i < OG; ++i) red[g & 1][wave][i][lane] = acc2[i];
    __syncthreads();
    // output fragment g * OG + wave is finished by this wave: the four hidden quarters in wave order
    f32x4 v = red[g & 1][0][wave][lane];
    v += red[g & 1][1][wave][lane];
    v += red[g & 1][2][wave][lane];
    v += red[g & 1][3][wave][lane];
    if (r0 + lr < R) *(float4*)(prow + (g * OG + wave) * 16) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// x' = res + (sum_s part[s] + bias) for the 8 rows of one sequence, then token mixing (MODE 0) or LayerNorm + token mean (MODE 1)
// on the workgroup's 64 channels.  NS = 0 (no slabs, bias may be null): x' = res — the first block of an iteration.
// 1024 threads: one float4 column of one row each, so a thread's NS slab loads (8 by default) are one round trip — hipcc keeps
// about ten loads in flight per wave whatever the source says, and a 256-thread version with 32 loads per thread took three.
template <int MODE, bool PART>
__global__ __launch_bounds__(1024) void k_pips_mix_reduce(const float* __restrict__ part, int NS, const float* __restrict__ bias,
                                                          const float* __restrict__ res, int R,
                                                          const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          float* __restrict__ out) {
  constexpr int S = 8, H = 4 * S, CH = 64;
  __shared__ __attribute__((aligned(16))) float xs[S][MD];
  __shared__ float psum[16], stat[S][2];
  __shared__ float sw1[H][S], sb1[H], sw2[S][H], sb2[S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq = blockIdx.x, c0 = blockIdx.y * CH;
  const int row = tid >> 7, c4 = (tid & 127) * 4;            // waves 2 row, 2 row + 1 hold row `row`
  const long idx = ((long)seq * S + row) * MD + c4;
  // the first 8 slabs lead the queue (vmcnt retires in order), then residual, bias and the token-mixing weights: one round trip
  float4 t0[8];
  if (PART) {
#pragma unroll
    for (int s = 0; s < 8; ++s) t0[s] = ld4(part + (long)s * R * MD + idx);
  }
  const float4 rv = ld4(res + idx);
  const float4 bv = PART ? ld4(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  // every small operand is loaded by every thread, unconditionally (a load under a branch costs its own round trip)
  const int cm = c0 + (MODE == 0 ? ((tid & 255) >> 2) : (tid & 63));
  const float gw = lnw[cm], gb = lnb[cm];
  if (MODE == 0) {
    const float a1 = w1[tid & 255], a2 = w2[tid & 255], a3 = b1[tid & 31], a4 = b2[tid & 7];
    if (tid < 256) sw1[tid >> 3][tid & 7] = a1, sw2[tid >> 5][tid & 31] = a2;
    if (tid < H) sb1[tid] = a3;
    if (tid < S) sb2[tid] = a4;
  }
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PART) {
#pragma unroll
    for (int s = 0; s < 8; ++s) v.x += t0[s].x, v.y += t0[s].y, v.z += t0[s].z, v.w += t0[s].w;
    const long slab = (long)R * MD;
    for (int s0 = 8; s0 < NS; s0 += 8) {
      float4 t[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) t[s] = ld4(part + (long)(s0 + s) * slab + idx);
#pragma unroll
      for (int s = 0; s < 8; ++s) v.x += t[s].x, v.y += t[s].y, v.z += t[s].z, v.w += t[s].w;
    }
  }
  v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
  v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
  *(float4*)&xs[row][c4] = v;
  // LayerNorm statistics of the 8 rows (two waves per row)
  {
    const float s = wave_sum((v.x + v.y) + (v.z + v.w));
    if (lane == 0) psum[wave] = s;
  }
  __syncthreads();
  const float mean = (psum[2 * row] + psum[2 * row + 1]) / (float)MD;
  __syncthreads();
  {
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float s = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    if (lane == 0) psum[wave] = s;
  }
  __syncthreads();
  if ((tid & 127) == 0) {
    stat[row][0] = mean;
    stat[row][1] = 1.0f / sqrtf((psum[2 * row] + psum[2 * row + 1]) / (float)MD + 1e-5f);
  }
  __syncthreads();
  if (MODE == 0) {
    if (tid >= 256) return;
    // thread -> (channel c, hidden-unit group og): 4 adjacent lanes share a channel and split the 32 hidden units
    const int c = cm, og = tid & 3;
    float xin[S], y[S];
#pragma unroll
    for (int t = 0; t < S; ++t) {
      xin[t] = xs[t][c];
      y[t] = (xin[t] - stat[t][0]) * stat[t][1] * gw + gb;
    }
    float acc[S];
#pragma unroll
    for (int t = 0; t < S; ++t) acc[t] = 0.f;
#pragma unroll
    for (int oo = 0; oo < H / 4; ++oo) {
      const int o = og * (H / 4) + oo;
      float a = sb1[o];
#pragma unroll
      for (int t = 0; t < S; ++t) a += sw1[o][t] * y[t];
      const float h = gelu_erf(a);
#pragma unroll
      for (int t = 0; t < S; ++t) acc[t] += sw2[t][o] * h;
    }
#pragma unroll
    for (int t = 0; t < S; ++t) {
      float a = acc[t];
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      acc[t] = a;
    }
#pragma unroll
    for (int t = 0; t < S; ++t)
      if ((t & 3) == og) out[((long)seq * S + t) * MD + c] = xin[t] + (acc[t] + sb2[t]);
  } else if (tid < CH) {
    const int c = cm;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < S; ++t) a += (xs[t][c] - stat[t][0]) * stat[t][1] * gw + gb;
    out[(long)seq * MD + c] = a / (float)S;
  }
}

int g_pips_mixer_fused = 1;     // sampt_pips_set_mixer: 1 = the two-launch blocks of this file, 0 = four launches per block
// workgroups a channel-MLP launch should reach: the exact-f32 kernel picks 8, 16 or 32 hidden slices per group of two chains, the
// split-fp16 kernel 16 (< 32) or 32 slices per 64 rows.  Default 16: beside the encoder's 30 persistent GEMM workgroups per XCD
// the chain then needs exactly the two CUs per XCD they leave (profiles/r6_c5_*, r6_c7_*: 113.9 fps against 111.8 with 32 / 28).
int g_pips_mixer_wgs = 16;
int g_pips_mixer_diag = 0;      // measurement only (SAMPT_PIPS_MIXER_DIAG=1): steady-state weight loads skipped — WRONG results, the
                                // launch's time without its weight stream

int pips_mix_slices(int nseq) {
  const int rgs = (nseq + 1) / 2;
  if (rgs * 8 >= g_pips_mixer_wgs) return 8;
  if (rgs * 16 >= g_pips_mixer_wgs) return 16;
  return 32;
}

int pips_mix_mlp(const float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                 float* part, int nseq, int NS, hipStream_t s) {
  if (nseq <= 0 || (NS != 8 && NS != 16 && NS != 32)) return SAMPT_ERR_ARG;
  const int R = nseq * 8, rgs = (nseq + 1) / 2;
  dim3 grid(rgs * NS), block(256);
#define MIXM(NFv, DIAGv) hipLaunchKernelGGL((k_pips_mix_mlp<NFv, DIAGv>), grid, block, 0, s, x, lnw, lnb, w1, b1, w2, part, R, NS)
  if (g_pips_mixer_diag) {
    if (NS == 8) MIXM(4, 1); else if (NS == 16) MIXM(2, 1); else MIXM(1, 1);
  } else {
    if (NS == 8) MIXM(4, 0); else if (NS == 16) MIXM(2, 0); else MIXM(1, 0);
  }
#undef MIXM
  SAMPT_CHECK_LAUNCH("pips_mix_mlp");
  return SAMPT_OK;
}

int pips_mix_reduce(const float* part, int NS, const float* bias, const float* res, int nseq, int mode, const float* lnw,
                    const float* lnb, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
                    hipStream_t s) {
  if (nseq <= 0 || NS < 0 || NS % 8 || (NS > 0 && (!part || !bias)) || !res || !out || res == out) return SAMPT_ERR_ARG;
  const int R = nseq * 8;
  dim3 grid(nseq, MD / 64), block(1024);
#define MIXR(MODEv, PARTv) \
  hipLaunchKernelGGL((k_pips_mix_reduce<MODEv, PARTv>), grid, block, 0, s, part, NS, bias, res, R, lnw, lnb, w1, b1, w2, b2, out)
  if (mode == 0) { if (NS) MIXR(0, true); else MIXR(0, false); }
  else { if (NS) MIXR(1, true); else MIXR(1, false); }
#undef MIXR
  SAMPT_CHECK_LAUNCH("pips_mix_reduce");
  return SAMPT_OK;
}

}  // namespace sampt
