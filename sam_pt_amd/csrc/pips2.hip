// PIPS++ point-update kernels (reference: sam_pt/point_tracker/pips_plus_plus/pips_plus_plus.py:263-342, 436-546).
//
// Row order of every [R = n*S][C] matrix is (point, frame): row = pt*S + s, the "B*N, S, C" order of the reference's
// DeltaBlock input (:510-516), so the 1-D convolutions over time are implicit GEMMs over an [n][S][1][C] NHWC image.
//   coords   : [S][n][2]  in stride-8 feature-map pixels
//   feats1/2/4 : [n][S][128] correlation templates (frame 0 / t-2 / t-4 features, :462-506)
//   x        : [n][S][720] = [corr1 196 | corr2 196 | corr4 196 | sincos 128 | flow 2 | pad 2]
#include "ops.h"

namespace sampt {

__device__ __forceinline__ float bilerp_feat(const float* __restrict__ fmap, int H, int W, int C, float x, float y,
                                             int c) {
  // bilinear_sample2d (utils/samp.py:6-80): clamped indices, weights from the un-clamped floor
  float x0f = floorf(x), y0f = floorf(y);
  float x1f = x0f + 1.f, y1f = y0f + 1.f;
  int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
  int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
  float w00 = (x1f - x) * (y1f - y), w01 = (x - x0f) * (y1f - y);
  float w10 = (x1f - x) * (y - y0f), w11 = (x - x0f) * (y - y0f);
  float v00 = fmap[((long)cy0 * W + cx0) * C + c], v01 = fmap[((long)cy0 * W + cx1) * C + c];
  float v10 = fmap[((long)cy1 * W + cx0) * C + c], v11 = fmap[((long)cy1 * W + cx1) * C + c];
  return w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11;
}

// coords = trajs0 / stride ; bak = coords[0] ; without feat_init all three templates = feature at frame 0 (:456-476)
__global__ __launch_bounds__(128) void k_pips2_init(const float* __restrict__ trajs0, const float* __restrict__ fmap, int H,
                                                    int W, const int* __restrict__ frame_idx, float stride, int S, int n,
                                                    int have_init, float* __restrict__ coords, float* __restrict__ bak,
                                                    float* __restrict__ f1, float* __restrict__ f2,
                                                    float* __restrict__ f4) {
  const int row = blockIdx.x, pt = row / S, s = row - pt * S, c = threadIdx.x;
  if (c < 2) {
    float v = trajs0[(s * n + pt) * 2 + c] / stride;
    coords[(s * n + pt) * 2 + c] = v;
    if (s == 0) bak[pt * 2 + c] = v;
  }
  if (have_init) return;
  const float x = trajs0[pt * 2] / stride, y = trajs0[pt * 2 + 1] / stride;       // frame 0 of the chunk
  const float v = bilerp_feat(fmap + (long)frame_idx[pt * S] * H * W * 128, H, W, 128, x, y, c);
  f1[(long)row * 128 + c] = v, f2[(long)row * 128 + c] = v, f4[(long)row * 128 + c] = v;
}

int pips2_init(const float* trajs0, const float* fmap, int H, int W, const int* frame_idx, float stride, int S, int n,
               int have_init, float* coords, float* bak, float* f1, float* f2, float* f4, hipStream_t s) {
  hipLaunchKernelGGL(k_pips2_init, dim3(n * S), dim3(128), 0, s, trajs0, fmap, H, W, frame_idx, stride, S, n, have_init,
                     coords, bak, f1, f2, f4);
  SAMPT_CHECK_LAUNCH("pips2_init");
  return SAMPT_OK;
}

// templates of iteration >= 1: feats_d[pt][s] = fmap[frame(max(s-d,0))] sampled at coords[max(s-d,0)][pt], d = 2, 4 (:490-506)
__global__ __launch_bounds__(128) void k_pips2_templates(const float* __restrict__ fmap, int H, int W,
                                                         const int* __restrict__ frame_idx,
                                                         const float* __restrict__ coords, int S, int n,
                                                         float* __restrict__ f2, float* __restrict__ f4) {
  const int row = blockIdx.x, pt = row / S, s = row - pt * S, c = threadIdx.x;
  const int d = blockIdx.y == 0 ? 2 : 4;
  const int src = max(s - d, 0);
  const float x = coords[(src * n + pt) * 2], y = coords[(src * n + pt) * 2 + 1];
  const float v = bilerp_feat(fmap + (long)frame_idx[pt * S + src] * H * W * 128, H, W, 128, x, y, c);
  (blockIdx.y == 0 ? f2 : f4)[(long)row * 128 + c] = v;
}

int pips2_templates(const float* fmap, int H, int W, const int* frame_idx, const float* coords, int S, int n, float* f2,
                    float* f4, hipStream_t s) {
  hipLaunchKernelGGL(k_pips2_templates, dim3(n * S, 2), dim3(128), 0, s, fmap, H, W, frame_idx, coords, S, n, f2, f4);
  SAMPT_CHECK_LAUNCH("pips2_templates");
  return SAMPT_OK;
}

// x[row][588:716] = posemb_sincos_2d_xy(flow, 128) ; x[row][716:718] = flow ; x[row][718:720] = 0  (misc.py:10-27, :512-519)
// flow[s] = coords[s+1] - coords[s], the last frame repeats the previous flow.  omega: device [32] = 1 / 10000^(k/31).
__global__ __launch_bounds__(128) void k_pips2_build_input(const float* __restrict__ coords,
                                                           const float* __restrict__ omega, int S, int n,
                                                           float* __restrict__ x, int ldx) {
  const int row = blockIdx.x, pt = row / S, s = row - pt * S, t = threadIdx.x;
  const int s0 = S > 1 ? min(s, S - 2) : 0, s1 = S > 1 ? s0 + 1 : 0;
  const float fx = coords[(s1 * n + pt) * 2] - coords[(s0 * n + pt) * 2];
  const float fy = coords[(s1 * n + pt) * 2 + 1] - coords[(s0 * n + pt) * 2 + 1];
  float* xr = x + (long)row * ldx + 588;
  const int q = t >> 5, k = t & 31;                 // q: 0 sin(x w)  1 cos(x w)  2 sin(y w)  3 cos(y w)
  const float a = (q < 2 ? fx : fy) * omega[k];
  xr[t] = (q & 1) ? cosf(a) : sinf(a);
  if (t < 4) xr[128 + t] = t == 0 ? fx : (t == 1 ? fy : 0.f);
}

int pips2_build_input(const float* coords, const float* omega, int S, int n, float* x, int ldx, hipStream_t s) {
  if (ldx != 720) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_pips2_build_input, dim3(n * S), dim3(128), 0, s, coords, omega, S, n, x, ldx);
  SAMPT_CHECK_LAUNCH("pips2_build_input");
  return SAMPT_OK;
}

// InstanceNorm1d over the S frames of each (point, channel) (no affine, eps 1e-5, biased variance) + ReLU
__global__ void k_instnorm1d_relu(const float* __restrict__ x, float* __restrict__ y, int S, int C) {
  const int pt = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float* xp = x + (long)pt * S * C + c;
  float sum = 0.f;
  for (int s = 0; s < S; ++s) sum += xp[(long)s * C];
  const float mean = sum / (float)S;
  float sq = 0.f;
  for (int s = 0; s < S; ++s) {
    float d = xp[(long)s * C] - mean;
    sq += d * d;
  }
  const float rstd = 1.0f / sqrtf(sq / (float)S + 1e-5f);
  float* yp = y + (long)pt * S * C + c;
  for (int s = 0; s < S; ++s) yp[(long)s * C] = fmaxf((xp[(long)s * C] - mean) * rstd, 0.f);
}

int instnorm1d_relu(const float* x, float* y, int n, int S, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_instnorm1d_relu, dim3(cdiv(C, 128), n), dim3(128), 0, s, x, y, S, C);
  SAMPT_CHECK_LAUNCH("instnorm1d_relu");
  return SAMPT_OK;
}

// ResidualBlock1d skip: out[row][c] += identity[row][c - ch1] for ch1 <= c < ch1 + cin (zero-padded channels, :96-104);
// relu != 0 applies DeltaBlock.final_relu to the sum
__global__ void k_add_chanpad(float* __restrict__ out, const float* __restrict__ identity, long rows, int cin, int cout,
                              int ch1, int relu) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cout) return;
  long row = i / cout;
  int c = (int)(i - row * cout) - ch1;
  float v = out[i];
  if (c >= 0 && c < cin) v += identity[row * cin + c];
  out[i] = relu ? fmaxf(v, 0.f) : v;
}

int add_chanpad(float* out, const float* identity, long rows, int cin, int cout, int relu, hipStream_t s) {
  const int ch1 = (cout - cin) / 2;
  hipLaunchKernelGGL(k_add_chanpad, dim3(cdiv(rows * cout, 256)), dim3(256), 0, s, out, identity, rows, cin, cout, ch1,
                     relu);
  SAMPT_CHECK_LAUNCH("add_chanpad");
  return SAMPT_OK;
}

// coords += delta ; coords[0] = bak (frame 0 is the locked target, :530-535); last != 0 also emits trajs = coords*stride
__global__ void k_pips2_apply_delta(const float* __restrict__ delta, const float* __restrict__ bak, float stride, int S,
                                    int n, int last, float* __restrict__ coords, float* __restrict__ trajs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * n * 2) return;
  const int c = i & 1, pt = (i >> 1) % n, s = (i >> 1) / n;
  float v = coords[i] + delta[((long)pt * S + s) * 2 + c];
  if (s == 0) v = bak[pt * 2 + c];
  coords[i] = v;
  if (last) trajs[i] = v * stride;
}

int pips2_apply_delta(const float* delta, const float* bak, float stride, int S, int n, int last, float* coords,
                      float* trajs, hipStream_t s) {
  hipLaunchKernelGGL(k_pips2_apply_delta, dim3(cdiv(S * n * 2, 256)), dim3(256), 0, s, delta, bak, stride, S, n, last,
                     coords, trajs);
  SAMPT_CHECK_LAUNCH("pips2_apply_delta");
  return SAMPT_OK;
}

}  // namespace sampt
