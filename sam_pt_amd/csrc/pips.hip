// PIPS point-update loop kernels (reference: sam_pt/point_tracker/pips/pips.py:290-407, 439-620).
//
// Data layout (all f32):
//   fmap pyramid  : level l = [frame][H_l][W_l][128]  NHWC — a pixel's 128 channels are one contiguous 512-B line,
//                   so the correlation footprint of a point (8 x 8 pixels) is 8 contiguous 4-KiB row segments.
//   ffeats        : [n][S][128]   (the mixer's "B*N, S, C" order, pips.py:529)
//   coords        : [S][n][2]     in stride-4 feature-map pixels
//   mixer input x : [n][S][ldx]   = [ffeat 128 | corr 4*49 | sincos 192 | flow,t 3 | pad]
#include "ops.h"

namespace sampt {

// ---------------------------------------------------------------------------------------------
// K7: bilinear_sample2d (utils/samp.py:6-80): clamped indices, weights from the un-clamped floor
// ---------------------------------------------------------------------------------------------
__global__ void k_pips_sample_feat(const float* __restrict__ fmap, int H, int W, int C,
                                   const int* __restrict__ frame_idx, const float* __restrict__ xy,
                                   float* __restrict__ out) {
  int pt = blockIdx.x, c = threadIdx.x;
  if (c >= C) return;
  if (frame_idx) fmap += (long)frame_idx[pt] * H * W * C;
  float x = xy[pt * 2], y = xy[pt * 2 + 1];
  float x0f = floorf(x), y0f = floorf(y);
  float x1f = x0f + 1.f, y1f = y0f + 1.f;
  int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
  int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
  float w00 = (x1f - x) * (y1f - y), w01 = (x - x0f) * (y1f - y);
  float w10 = (x1f - x) * (y - y0f), w11 = (x - x0f) * (y - y0f);
  float v00 = fmap[((long)cy0 * W + cx0) * C + c], v01 = fmap[((long)cy0 * W + cx1) * C + c];
  float v10 = fmap[((long)cy1 * W + cx0) * C + c], v11 = fmap[((long)cy1 * W + cx1) * C + c];
  out[pt * C + c] = w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11;
}

int pips_sample_feat(const float* fmap, int H, int W, int C, const int* frame_idx, const float* xy, int n, float* out,
                     hipStream_t s) {
  if (n <= 0 || C > 1024) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_pips_sample_feat, dim3(n), dim3(C), 0, s, fmap, H, W, C, frame_idx, xy, out);
  SAMPT_CHECK_LAUNCH("pips_sample_feat");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// K8+K9 fused: local correlation + 7x7 bilinear window (CorrBlock.corr + CorrBlock.sample)
// One wave per (frame s, point, level).  The reference materialises <ffeat, fmap> over the WHOLE map and then
// grid_samples 49 taps; all taps share one 8x8 pixel footprint, so only those 64 dot products are computed:
// 64 px * 512 B = 32 KiB of compulsory HBM traffic per unit instead of streaming the pyramid.
// Lane (fx, q): pixel column fx of the footprint, channel chunk q (16 channels); 8 footprint rows are walked
// with 4 independent float4 loads per lane per row (each 8-lane group reads one contiguous 512-B pixel).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pips_corr_sample(PyramidLevels pyr, const int* __restrict__ frame_idx, int S,
                                                          int n, const float* __restrict__ ffeats,
                                                          const float* __restrict__ coords, float* __restrict__ x,
                                                          int ldx, int xoff, const float* __restrict__ times) {
  constexpr int C = 128;
  __shared__ float cs[8][8];
  const int lane = threadIdx.x;
  const int unit = blockIdx.x, lvl = blockIdx.y;
  const int s = unit / n, pt = unit - s * n;
  const int H = pyr.H[lvl], W = pyr.W[lvl];
  const float* fmap = pyr.base[lvl] + (long)frame_idx[pt * S + s] * H * W * C;  // per-point window frames
  const float scale = (float)(1 << lvl);
  const float cx = coords[(s * n + pt) * 2] / scale, cy = coords[(s * n + pt) * 2 + 1] / scale;
  const int bx = (int)floorf(cx) - 3, by = (int)floorf(cy) - 3;

  const int fx = lane >> 3, q = lane & 7;
  const float4* f4 = (const float4*)(ffeats + ((long)pt * S + s) * C + q * 16);
  const float4 a0 = f4[0], a1 = f4[1], a2 = f4[2], a3 = f4[3];
  const int px = bx + fx;
  const bool xin = px >= 0 && px < W;
#pragma unroll
  for (int fy = 0; fy < 8; ++fy) {
    int py = by + fy;
    float d = 0.f;
    if (xin && py >= 0 && py < H) {
      const float4* p = (const float4*)(fmap + ((long)py * W + px) * C + q * 16);
      float4 b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
      d = a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
      d += a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
      d += a2.x * b2.x + a2.y * b2.y + a2.z * b2.z + a2.w * b2.w;
      d += a3.x * b3.x + a3.y * b3.y + a3.z * b3.z + a3.w * b3.w;
    }
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if (q == 0) cs[fy][fx] = d / sqrtf((float)C);  // pips.py:406; out-of-map pixels stay 0 (zeros padding)
  }
  __syncthreads();
  if (lane < 49) {
    const int i = lane / 7, j = lane - i * 7;
    // tap (i,j) samples at (x + (i-3), y + (j-3)): the reference adds (dy_i, dx_j) to (x, y) (pips.py:378-384)
    float posx = cx + (float)(i - 3), posy = cy + (float)(j - 3);
    // bilinear_sampler normalisation (pips.py:324-326) and grid_sample's align_corners=True un-normalisation
    float gx = 2.f * posx / (float)(W - 1) - 1.f, gy = 2.f * posy / (float)(H - 1) - 1.f;
    float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    float xw = floorf(ix), yn = floorf(iy);
    float w = ix - xw, e = 1.f - w, nn = iy - yn, ss = 1.f - nn;
    int rx = (int)xw - bx, ry = (int)yn - by;
    int rx0 = min(max(rx, 0), 7), rx1 = min(max(rx + 1, 0), 7);
    int ry0 = min(max(ry, 0), 7), ry1 = min(max(ry + 1, 0), 7);
    float v = cs[ry0][rx0] * (ss * e) + cs[ry0][rx1] * (ss * w) + cs[ry1][rx0] * (nn * e) + cs[ry1][rx1] * (nn * w);
    x[((long)pt * S + s) * ldx + xoff + lvl * 49 + lane] = v;
  }
  // the level-0 unit of a row also writes the rest of the mixer input (k_pips_build_input's arithmetic, one launch less)
  if (times && lvl == 0) {
    float* xr = x + ((long)pt * S + s) * ldx;
    const float* fr = ffeats + ((long)pt * S + s) * C;
    xr[lane] = fr[lane], xr[lane + 64] = fr[lane + 64];
    const float fx = coords[(s * n + pt) * 2] - coords[pt * 2];
    const float fy = coords[(s * n + pt) * 2 + 1] - coords[pt * 2 + 1];
    const float tz = times[s];
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
      const int k = lane;
      const float v = axis == 0 ? fx : (axis == 1 ? fy : tz);
      const float div = (float)(k & ~1) * (1000.0f / 64.0f);
      const float a = v * div;
      xr[324 + axis * 64 + k] = (k & 1) ? cosf(a) : sinf(a);
    }
    if (lane < 3) xr[324 + 192 + lane] = lane == 0 ? fx : (lane == 1 ? fy : tz);
    else if (lane < 3 + (ldx - 519)) xr[324 + 192 + lane] = 0.f;  // K padding
  }
}

int pips_corr_sample(const PyramidLevels& pyr, const int* frame_idx, int S, int n, int C, const float* ffeats,
                     const float* coords, float* x, int ldx, int xoff, hipStream_t s, const float* times) {
  if (C != 128 || n <= 0 || S <= 0) return SAMPT_ERR_ARG;
  if (times && (xoff != 128 || ldx < 519 || ldx > 519 + 61)) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_pips_corr_sample, dim3(S * n, 4), dim3(64), 0, s, pyr, frame_idx, S, n, ffeats, coords, x, ldx,
                     xoff, times);
  SAMPT_CHECK_LAUNCH("pips_corr_sample");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// K11 + concat: x[:, :, 0:128] = ffeats ; x[:, :, 324:519] = [sincos(fx), sincos(fy), sincos(t), fx, fy, t]
// (utils/misc.py:30-55 with C=64; pips.py:525-530).  One workgroup (256 threads) per (point, frame) row.
// ---------------------------------------------------------------------------------------------
__global__ void k_pips_build_input(const float* __restrict__ ffeats, const float* __restrict__ coords,
                                   const float* __restrict__ times, int S, int n, float* __restrict__ x, int ldx) {
  const int row = blockIdx.x;  // pt*S + s
  const int pt = row / S, s = row - pt * S;
  const int t = threadIdx.x;
  float* xr = x + (long)row * ldx;
  if (t < 128) xr[t] = ffeats[(long)row * 128 + t];
  float fx = coords[(s * n + pt) * 2] - coords[pt * 2];
  float fy = coords[(s * n + pt) * 2 + 1] - coords[pt * 2 + 1];
  float tz = times[s];
  if (t < 192) {
    int axis = t / 64, k = t - axis * 64;
    float v = axis == 0 ? fx : (axis == 1 ? fy : tz);
    float div = (float)(k & ~1) * (1000.0f / 64.0f);
    float a = v * div;
    xr[324 + t] = (k & 1) ? cosf(a) : sinf(a);
  } else if (t < 195) {
    xr[324 + t] = t == 192 ? fx : (t == 193 ? fy : tz);
  } else if (t < 195 + (ldx - 519)) {
    xr[324 + t] = 0.f;  // K padding
  }
}

int pips_build_input(const float* ffeats, const float* coords, const float* times, int S, int n, float* x, int ldx,
                          hipStream_t s) {
  if (ldx < 519 || ldx > 519 + 61) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_pips_build_input, dim3(n * S), dim3(256), 0, s, ffeats, coords, times, S, n, x, ldx);
  SAMPT_CHECK_LAUNCH("pips_build_input");
  return SAMPT_OK;
}

__global__ void k_pips_init_state(const float* __restrict__ xys, const float* __restrict__ feat_init, float stride,
                                  int S, int n, float* __restrict__ coords, float* __restrict__ coords0,
                                  float* __restrict__ ffeats) {
  const int row = blockIdx.x, pt = row / S, s = row - pt * S, c = threadIdx.x;
  ffeats[(long)row * 128 + c] = feat_init[pt * 128 + c];
  if (c < 2) {
    float v = xys[pt * 2 + c] / stride;  // pips.py:458
    coords[(s * n + pt) * 2 + c] = v;
    if (s == 0) coords0[pt * 2 + c] = v;
  }
}

int pips_init_state(const float* xys, const float* feat_init, float stride, int S, int n, float* coords,
                    float* coords0, float* ffeats, hipStream_t s) {
  hipLaunchKernelGGL(k_pips_init_state, dim3(n * S), dim3(128), 0, s, xys, feat_init, stride, S, n, coords, coords0,
                     ffeats);
  SAMPT_CHECK_LAUNCH("pips_init_state");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// K12a: token-mixing block  x = x + W2 . gelu(W1 . LN(x) + b1) + b2  over the S=8 tokens (Conv1d k=1)
// One workgroup per sequence; the 8 x 512 activations live in LDS.
// ---------------------------------------------------------------------------------------------
template <int S, int D>
__global__ __launch_bounds__(256) void k_pips_token_mix(const float* __restrict__ x, float* __restrict__ xo,
                                                        const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2) {
  // grid = (sequence, D/64 channel chunks).  Every workgroup recomputes the LayerNorm statistics of the 8 tokens
  // (16 KiB of L2-resident reads) and then mixes its own 64 channels; out of place, so chunks never race.
  constexpr int H = 4 * S, CH = 64;
  __shared__ float stat[S][2];
  __shared__ float sw1[H][S], sb1[H], sw2[S][H], sb2[S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xs = x + (long)blockIdx.x * S * D;
  float* xos = xo + (long)blockIdx.x * S * D;
  const int c0 = blockIdx.y * CH;
  auto ld = [&](int tok, int ch) -> float { return xs[tok * D + ch]; };
  for (int i = tid; i < H * S; i += 256) {
    ((float*)sw1)[i] = w1[i];
    ((float*)sw2)[i] = w2[i];
  }
  if (tid < H) sb1[tid] = b1[tid];
  if (tid < S) sb2[tid] = b2[tid];
  for (int tok = wave; tok < S; tok += 4) {
    float v[D / 64];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < D / 64; ++i) {
      v[i] = ld(tok, lane + 64 * i);
      sum += v[i];
    }
    float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < D / 64; ++i) {
      float d = v[i] - mean;
      sq += d * d;
    }
    float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
    if (lane == 0) stat[tok][0] = mean, stat[tok][1] = rstd;
  }
  __syncthreads();
  // thread -> (channel c, hidden-unit group og): 4 adjacent lanes share a channel and split the 32 hidden units
  const int c = c0 + (tid >> 2), og = tid & 3;
  const float gw = lnw[c], gb = lnb[c];
  float xin[S], y[S];
#pragma unroll
  for (int t = 0; t < S; ++t) {
    xin[t] = ld(t, c);
    y[t] = (xin[t] - stat[t][0]) * stat[t][1] * gw + gb;
  }
  float part[S];
#pragma unroll
  for (int t = 0; t < S; ++t) part[t] = 0.f;
#pragma unroll
  for (int oo = 0; oo < H / 4; ++oo) {
    const int o = og * (H / 4) + oo;
    float a = sb1[o];
#pragma unroll
    for (int t = 0; t < S; ++t) a += sw1[o][t] * y[t];
    const float h = gelu_erf(a);
#pragma unroll
    for (int t = 0; t < S; ++t) part[t] += sw2[t][o] * h;
  }
#pragma unroll
  for (int t = 0; t < S; ++t) {
    float a = part[t];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    part[t] = a;
  }
#pragma unroll
  for (int t = 0; t < S; ++t)
    if ((t & 3) == og) xos[t * D + c] = xin[t] + (part[t] + sb2[t]);
}

int pips_token_mix(const float* x, float* xo, const float* lnw, const float* lnb, const float* w1, const float* b1,
                   const float* w2, const float* b2, int nseq, int S, int D, hipStream_t s) {
  if (S != 8 || D != 512 || nseq <= 0 || x == xo) return SAMPT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((k_pips_token_mix<8, 512>), dim3(nseq, D / 64), dim3(256), 0, s, x, xo, lnw, lnb, w1, b1, w2, b2);
  SAMPT_CHECK_LAUNCH("pips_token_mix");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// K12b: final LayerNorm + mean over the S tokens (pips.py:125-126)
// ---------------------------------------------------------------------------------------------
template <int S, int D>
__global__ __launch_bounds__(256) void k_pips_ln_mean(const float* __restrict__ x, const float* __restrict__ lnw,
                                                      const float* __restrict__ lnb, float* __restrict__ out) {
  __shared__ float ys[S][D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xs = x + (long)blockIdx.x * S * D;
  for (int tok = wave; tok < S; tok += 4) {
    float v[D / 64];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < D / 64; ++i) {
      v[i] = xs[tok * D + lane + 64 * i];
      sum += v[i];
    }
    float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < D / 64; ++i) {
      float d = v[i] - mean;
      sq += d * d;
    }
    float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < D / 64; ++i) {
      int c = lane + 64 * i;
      ys[tok][c] = (v[i] - mean) * rstd * lnw[c] + lnb[c];
    }
  }
  __syncthreads();
  for (int c = tid; c < D; c += 256) {
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < S; ++t) a += ys[t][c];
    out[(long)blockIdx.x * D + c] = a / (float)S;
  }
}

int pips_ln_mean(const float* x, const float* lnw, const float* lnb, float* out, int nseq, int S, int D, hipStream_t s) {
  if (S != 8 || D != 512 || nseq <= 0) return SAMPT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((k_pips_ln_mean<8, 512>), dim3(nseq), dim3(256), 0, s, x, lnw, lnb, out);
  SAMPT_CHECK_LAUNCH("pips_ln_mean");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// K13: ffeat += gelu(Linear(GroupNorm(1,128)(dfeat)));  coords += dxy;  coords[0] locked   (pips.py:536-544)
// One workgroup (128 threads) per (point, frame) row.  up_wT is the Linear weight transposed to [in][out].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_pips_update(const float* __restrict__ delta, const float* __restrict__ gn_w,
                                                     const float* __restrict__ gn_b, const float* __restrict__ up_wT,
                                                     const float* __restrict__ up_b, float* __restrict__ ffeats,
                                                     float* __restrict__ coords, const float* __restrict__ coords0,
                                                     int S, int n) {
  constexpr int C = 128;
  __shared__ float g[C];
  __shared__ float red[4];
  const int row = blockIdx.x, pt = row / S, s = row - pt * S;
  const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
  const float* d = delta + (long)row * (C + 2);
  float v = d[2 + c];
  float sum = wave_sum(v);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  float mean = (red[0] + red[1]) / (float)C;
  float dv = v - mean;
  float sq = wave_sum(dv * dv);
  if (lane == 0) red[2 + wave] = sq;
  __syncthreads();
  float rstd = 1.0f / sqrtf((red[2] + red[3]) / (float)C + 1e-5f);
  g[c] = dv * rstd * gn_w[c] + gn_b[c];
  __syncthreads();
  float a = up_b[c];
#pragma unroll 8
  for (int k = 0; k < C; ++k) a += up_wT[k * C + c] * g[k];
  ffeats[(long)row * C + c] += gelu_erf(a);
  if (c < 2) {
    int ci = (s * n + pt) * 2 + c;
    coords[ci] = (s == 0 && coords0) ? coords0[pt * 2 + c] : coords[ci] + d[c];   // coords0 null: no lock (CoTracker)
  }
}

int pips_update(const float* delta, const float* gn_w, const float* gn_b, const float* up_wT, const float* up_b,
                float* ffeats, float* coords, const float* coords0, int S, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_pips_update, dim3(n * S), dim3(128), 0, s, delta, gn_w, gn_b, up_wT, up_b, ffeats, coords,
                     coords0, S, n);
  SAMPT_CHECK_LAUNCH("pips_update");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// K14: visibility head + sigmoid (pips.py:568, pips/tracker.py:102) and trajectories in image pixels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pips_finalize(const float* __restrict__ ffeats, const float* __restrict__ vis_w,
                                                       const float* __restrict__ vis_b, const float* __restrict__ coords,
                                                       float stride, int S, int n, float* __restrict__ traj,
                                                       float* __restrict__ vis) {
  const int row = blockIdx.x, pt = row / S, s = row - pt * S, lane = threadIdx.x;
  const float* f = ffeats + (long)row * 128;
  float a = f[lane] * vis_w[lane] + f[lane + 64] * vis_w[lane + 64];
  a = wave_sum(a);
  if (lane == 0) {
    float logit = a + vis_b[0];
    vis[s * n + pt] = 1.0f / (1.0f + expf(-logit));
  }
  if (lane < 2) traj[(s * n + pt) * 2 + lane] = coords[(s * n + pt) * 2 + lane] * stride;
}

int pips_finalize(const float* ffeats, const float* vis_w, const float* vis_b, const float* coords, float stride,
                  int S, int n, float* traj, float* vis, hipStream_t s) {
  hipLaunchKernelGGL(k_pips_finalize, dim3(n * S), dim3(64), 0, s, ffeats, vis_w, vis_b, coords, stride, S, n, traj, vis);
  SAMPT_CHECK_LAUNCH("pips_finalize");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// Chained windows of PipsPointTracker._forward (pips/tracker.py:42-153) with the bookkeeping ON THE DEVICE: every chain
// (one point in one temporal direction) owns an anchor frame `cur`; a ROUND runs one 8-frame window for every chain at
// once (the mixer treats points independently, pips.py:525-532).  Chain time axis d = 0..T-1; a flipped chain reads
// pyramid frame T-1-d (tracker.py:162-167).  traj [T][n][2] px, vis [T][n] = sigmoid(logit) (0 where never written).
// ---------------------------------------------------------------------------------------------
// state reset + query frame entries (tracker.py:57-63): one thread per chain
__global__ void k_pips_chain_init(const float* __restrict__ q /*[n][3] = (t, x, y)*/, int n, int* __restrict__ cur,
                                  float* __restrict__ traj, float* __restrict__ vis) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t0 = (int)q[i * 3];
  cur[i] = t0;
  traj[((long)t0 * n + i) * 2] = q[i * 3 + 1];
  traj[((long)t0 * n + i) * 2 + 1] = q[i * 3 + 2];
  vis[(long)t0 * n + i] = 1.0f;
}

// window frames (the last frame repeated when the clip ends inside the window, tracker.py:73-78) and anchor positions of
// every chain for this round; finished chains keep a valid dummy window (their results are never written back)
__global__ void k_pips_round_begin(const int* __restrict__ cur, const unsigned char* __restrict__ flip, const float* __restrict__ traj,
                                   int T, int n, int S, int* __restrict__ fidx /*[n][S]*/, float* __restrict__ xys /*[n][2]*/,
                                   float* __restrict__ xy_feat /*[n][2] or null: anchor / stride (first round only)*/,
                                   int* __restrict__ f0 /*[n] or null: pyramid frame of the anchor*/, float stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int f = cur[i];
  if (f > T - 1) f = T - 1;
  const int hi = min(T - f, S);
  for (int s = 0; s < S; ++s) {
    const int w = min(f + s, f + hi - 1);
    fidx[i * S + s] = flip[i] ? T - 1 - w : w;
  }
  const float x = traj[((long)f * n + i) * 2], y = traj[((long)f * n + i) * 2 + 1];
  xys[i * 2] = x, xys[i * 2 + 1] = y;
  if (xy_feat) xy_feat[i * 2] = x / stride, xy_feat[i * 2 + 1] = y / stride, f0[i] = flip[i] ? T - 1 - f : f;
}

// write frames 1 .. hi-1 of every active chain's window (tracker.py:104-109), then link: the next anchor is the latest
// frame of the window whose visibility exceeds the threshold, the threshold decaying by 0.02 per sweep (tracker.py:111-148;
// float32 arithmetic like the reference's torch code).  n_active[0] = chains that still have frames to track.
__global__ void k_pips_round_end(int* __restrict__ cur, const float* __restrict__ tr /*[S][n][2]*/, const float* __restrict__ vi /*[S][n]*/,
                                 int T, int n, int S, float thr0, float* __restrict__ traj, float* __restrict__ vis,
                                 int* __restrict__ n_active) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int still = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int f = cur[i];
    if (f >= T - 1) continue;                      // tracker.py:67: anchors range over n_frames - 1
    const int hi = min(T - f, S);
    for (int s = 1; s < hi; ++s) {
      vis[(long)(f + s) * n + i] = vi[s * n + i];
      traj[((long)(f + s) * n + i) * 2] = tr[(s * n + i) * 2];
      traj[((long)(f + s) * n + i) * 2 + 1] = tr[(s * n + i) * 2 + 1];
    }
    float thr = thr0;
    const int earliest = f + 1, last = f + hi - 1;
    int nxt = last;
    while (vis[(long)nxt * n + i] <= thr) {
      nxt -= 1;
      if (nxt < earliest) thr = thr - 0.02f, nxt = last;
    }
    cur[i] = nxt;
    still += nxt < T - 1;
  }
  if (still) atomicAdd(&cnt, still);
  __syncthreads();
  if (threadIdx.x == 0) n_active[0] = cnt;
}

int pips_chain_init(const float* q, int n, int T, int* cur, float* traj, float* vis, hipStream_t s) {
  if (hipMemsetAsync(traj, 0, (size_t)T * n * 2 * sizeof(float), s) != hipSuccess) return SAMPT_ERR_HIP;
  if (hipMemsetAsync(vis, 0, (size_t)T * n * sizeof(float), s) != hipSuccess) return SAMPT_ERR_HIP;
  hipLaunchKernelGGL(k_pips_chain_init, dim3(cdiv(n, 64)), dim3(64), 0, s, q, n, cur, traj, vis);
  SAMPT_CHECK_LAUNCH("pips_chain_init");
  return SAMPT_OK;
}

int pips_round_begin(const int* cur, const unsigned char* flip, const float* traj, int T, int n, int S, int* fidx, float* xys,
                     float* xy_feat, int* f0, float stride, hipStream_t s) {
  hipLaunchKernelGGL(k_pips_round_begin, dim3(cdiv(n, 64)), dim3(64), 0, s, cur, flip, traj, T, n, S, fidx, xys, xy_feat, f0,
                     stride);
  SAMPT_CHECK_LAUNCH("pips_round_begin");
  return SAMPT_OK;
}

int pips_round_end(int* cur, const float* tr, const float* vi, int T, int n, int S, float thr0, float* traj, float* vis,
                   int* n_active, hipStream_t s) {
  hipLaunchKernelGGL(k_pips_round_end, dim3(1), dim3(256), 0, s, cur, tr, vi, T, n, S, thr0, traj, vis, n_active);
  SAMPT_CHECK_LAUNCH("pips_round_end");
  return SAMPT_OK;
}

}  // namespace sampt
