// CoTracker (v1) window kernels — SURVEY.md §8 row a13, Appendix A-6.  The model is third-party
// (facebookresearch/co-tracker @ 4f297a9: CoTracker.forward / forward_iteration, UpdateFormer); the adapter that calls it is
// sam_pt/point_tracker/cotracker/tracker.py:27-170.  The correlation sampler, the feature/coordinate update and the
// encoder are the PIPS kernels (upstream copied those modules from PIPS); this file adds what differs.
//
// Layouts (f32, one window of S = 8 frames, na = points active in it, rows r = pt*S + s):
//   coords [S][na][2] feature-map px      ffeats [na][S][128]      x [na][S][456] transformer input
//   token stream h [na*S][384]; time attention = tokens of one point, space attention = tokens of one frame.
#include "ops.h"

namespace sampt {

// once per track() call: per-point query frame / position in feature-map pixels, and the "never written" output values
// (trajectory 0 — what the adapter's `== 0` back-fill keys on, tracker.py:166 — and sigmoid(0) = 0.5)
__global__ void k_cot_prepare(const float* __restrict__ qxy, const int* __restrict__ qt, const int* __restrict__ frame_map,
                              float stride, int n, int T, float* __restrict__ xy0, int* __restrict__ fidx_pt,
                              float* __restrict__ traj_out, float* __restrict__ vis_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    xy0[2 * i] = qxy[2 * i] / stride, xy0[2 * i + 1] = qxy[2 * i + 1] / stride;
    fidx_pt[i] = frame_map[qt[i]];
  }
  if (i < (long)T * n) {
    traj_out[2 * i] = 0.f, traj_out[2 * i + 1] = 0.f;
    vis_out[i] = 0.5f;
  }
}

int cot_prepare(const float* qxy, const int* qt, const int* frame_map, float stride, int n, int T, float* xy0, int* fidx_pt,
                float* traj_out, float* vis_out, hipStream_t s) {
  hipLaunchKernelGGL(k_cot_prepare, dim3(cdiv((long)T * n, 256)), dim3(256), 0, s, qxy, qt, frame_map, stride, n, T, xy0,
                     fidx_pt, traj_out, vis_out);
  SAMPT_CHECK_LAUNCH("cot_prepare");
  return SAMPT_OK;
}

// Window state (CoTracker.forward): points [0, prev) were in the previous window (started 4 frames earlier) and carry its
// second half over — frames 0..3 <- previous frames 4..7, frames 4..7 <- previous frame 7, for coordinates and visibility
// logits; points [prev, na) start here: query position on every frame, visibility logit 10.  track_mask = frames at or after
// the query frame that no earlier window has written; padded tail frames (s >= S_local) are masked out.
__global__ __launch_bounds__(128) void k_cot_window_init(int ind, int S_local, int prev, int na, int S,
                                                         const int* __restrict__ qt, const float* __restrict__ xy0,
                                                         const int* __restrict__ frame_map,
                                                         const float* __restrict__ coords_prev,
                                                         const float* __restrict__ vis_prev,
                                                         const float* __restrict__ feat_init, float* __restrict__ coords,
                                                         float* __restrict__ visin, float* __restrict__ mask,
                                                         int* __restrict__ fidx, float* __restrict__ ffeats) {
  const int row = blockIdx.x, pt = row / S, s = row - pt * S, c = threadIdx.x;
  ffeats[(long)row * 128 + c] = feat_init[pt * 128 + c];
  const bool old = pt < prev;
  const int src = min(s + S / 2, S - 1);
  if (c < 2) coords[(s * na + pt) * 2 + c] = old ? coords_prev[(src * prev + pt) * 2 + c] : xy0[pt * 2 + c];
  if (c == 2) visin[s * na + pt] = old ? vis_prev[src * prev + pt] : 10.0f;
  if (c == 3) mask[s * na + pt] = (s < S_local && (old ? s >= S / 2 : ind + s >= qt[pt])) ? 1.0f : 0.0f;
  if (c == 4) fidx[pt * S + s] = frame_map[ind + min(s, S_local - 1)];
}

int cot_window_init(int ind, int S_local, int prev, int na, int S, const int* qt, const float* xy0, const int* frame_map,
                    const float* coords_prev, const float* vis_prev, const float* feat_init, float* coords, float* visin,
                    float* mask, int* fidx, float* ffeats, hipStream_t s) {
  hipLaunchKernelGGL(k_cot_window_init, dim3(na * S), dim3(128), 0, s, ind, S_local, prev, na, S, qt, xy0, frame_map,
                     coords_prev, vis_prev, feat_init, coords, visin, mask, fidx, ffeats);
  SAMPT_CHECK_LAUNCH("cot_window_init");
  return SAMPT_OK;
}

// sample_pos_embed: the 2-D sin/cos grid embedding (first E/2 channels a function of the column, last E/2 of the row)
// sampled with bilinear_sample2d (clamped indices, weights from the unclamped floor) at the window's first-frame position.
// pos_x [W][E/2], pos_y [H][E/2] are the two 1-D tables; the 4-tap sum keeps bilinear_sample2d's order of operations.
__global__ void k_cot_pos_embed(const float* __restrict__ coords, const float* __restrict__ pos_x,
                                const float* __restrict__ pos_y, int H, int W, int E, float* __restrict__ pos) {
  const int pt = blockIdx.x;
  const float x = coords[pt * 2], y = coords[pt * 2 + 1];
  const float x0f = floorf(x), y0f = floorf(y), x1f = x0f + 1.f, y1f = y0f + 1.f;
  const int cx0 = min(max((int)x0f, 0), W - 1), cx1 = min(max((int)x0f + 1, 0), W - 1);
  const int cy0 = min(max((int)y0f, 0), H - 1), cy1 = min(max((int)y0f + 1, 0), H - 1);
  const float w00 = (x1f - x) * (y1f - y), w01 = (x - x0f) * (y1f - y), w10 = (x1f - x) * (y - y0f), w11 = (x - x0f) * (y - y0f);
  const int half = E / 2;
  for (int c = threadIdx.x; c < E; c += blockDim.x) {
    float v00, v01, v10, v11;
    if (c < half) {
      v00 = v10 = pos_x[cx0 * half + c], v01 = v11 = pos_x[cx1 * half + c];
    } else {
      v00 = v01 = pos_y[cy0 * half + c - half], v10 = v11 = pos_y[cy1 * half + c - half];
    }
    pos[(long)pt * E + c] = w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11;
  }
}

int cot_pos_embed(const float* coords, const float* pos_x, const float* pos_y, int H, int W, int E, int na, float* pos,
                  hipStream_t s) {
  hipLaunchKernelGGL(k_cot_pos_embed, dim3(na), dim3(256), 0, s, coords, pos_x, pos_y, H, W, E, pos);
  SAMPT_CHECK_LAUNCH("cot_pos_embed");
  return SAMPT_OK;
}

// transformer input (forward_iteration): x = [flow embedding 130 | correlation 196 (already written by the correlation
// sampler) | track feature 128 | track mask, visibility logit] + position embedding + time embedding
__global__ __launch_bounds__(256) void k_cot_build_input(const float* __restrict__ ffeats, const float* __restrict__ coords,
                                                         const float* __restrict__ visin, const float* __restrict__ mask,
                                                         const float* __restrict__ pos, const float* __restrict__ times,
                                                         int S, int na, float* __restrict__ x) {
  constexpr int E = 456;
  const int row = blockIdx.x, pt = row / S, s = row - pt * S;
  float* xr = x + (long)row * E;
  const float fx = coords[(s * na + pt) * 2] - coords[pt * 2];
  const float fy = coords[(s * na + pt) * 2 + 1] - coords[pt * 2 + 1];
  for (int c = threadIdx.x; c < E; c += 256) {
    float v;
    if (c < 2) {
      v = c == 0 ? fx : fy;
    } else if (c < 130) {                       // get_2d_embedding(C = 64): sin/cos interleaved, frequencies k/2 * 1000/64 * 2
      const int k = (c - 2) & 63;
      const float a = ((c - 2) < 64 ? fx : fy) * ((float)(k & ~1) * (1000.0f / 64.0f));
      v = (k & 1) ? cosf(a) : sinf(a);
    } else if (c < 326) {
      v = xr[c];
    } else if (c < 454) {
      v = ffeats[(long)row * 128 + c - 326];
    } else {
      v = c == 454 ? mask[s * na + pt] : visin[s * na + pt];
    }
    xr[c] = (v + pos[(long)pt * E + c]) + times[s * E + c];
  }
}

int cot_build_input(const float* ffeats, const float* coords, const float* visin, const float* mask, const float* pos,
                    const float* times, int S, int na, float* x, hipStream_t s) {
  hipLaunchKernelGGL(k_cot_build_input, dim3(na * S), dim3(256), 0, s, ffeats, coords, visin, mask, pos, times, S, na, x);
  SAMPT_CHECK_LAUNCH("cot_build_input");
  return SAMPT_OK;
}

// Multi-head attention over short token groups straight from the packed qkv rows (timm Attention): one wave per
// (query, head); lanes own keys lane, lane + 64, ...; scores stay in registers.  Token t of group b is row b*bs + t*ts of
// qkv [rows][3*heads*HD] / out [rows][heads*HD]: time attention bs = S, ts = 1; space attention bs = 1, ts = S.
template <int HD, int KPT>
__global__ __launch_bounds__(256) void k_cot_attention(const float* __restrict__ qkv, float* __restrict__ out, int L, int bs,
                                                       int ts, int heads) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  if (qi >= L) return;
  const int ld = 3 * heads * HD, D = heads * HD;
  const float* base = qkv + (long)b * bs * ld;
  const float4* qp = (const float4*)(base + (long)qi * ts * ld + h * HD);
  const float scale = 1.0f / sqrtf((float)HD);
  float q[HD];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    float4 t = qp[c];
    q[4 * c] = t.x * scale, q[4 * c + 1] = t.y * scale, q[4 * c + 2] = t.z * scale, q[4 * c + 3] = t.w * scale;
  }
  float sc[KPT];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int key = lane + 64 * i;
    sc[i] = -INFINITY;
    if (key < L) {
      const float4* kp = (const float4*)(base + (long)key * ts * ld + D + h * HD);
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        float4 t = kp[c];
        a += q[4 * c] * t.x + q[4 * c + 1] * t.y + q[4 * c + 2] * t.z + q[4 * c + 3] * t.w;
      }
      sc[i] = a;
      m = fmaxf(m, a);
    }
  }
  m = wave_max(m);
  float acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int key = lane + 64 * i;
    if (key < L) {
      const float p = expf(sc[i] - m);
      sum += p;
      const float4* vp = (const float4*)(base + (long)key * ts * ld + 2 * D + h * HD);
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        float4 t = vp[c];
        acc[4 * c] += p * t.x, acc[4 * c + 1] += p * t.y, acc[4 * c + 2] += p * t.z, acc[4 * c + 3] += p * t.w;
      }
    }
  }
  sum = wave_sum(sum);
  float* op = out + ((long)b * bs + (long)qi * ts) * D + h * HD;
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    const float a = wave_sum(acc[c]);
    if (lane == (c & 63)) op[c] = a / sum;
  }
}

int cot_attention(const float* qkv, float* out, int nbatch, int L, int bs, int ts, int heads, int hd, hipStream_t s) {
  if (hd != 48 || L <= 0 || nbatch <= 0) return SAMPT_ERR_UNSUPPORTED;
  dim3 grid(cdiv(L, 4), heads, nbatch);
  if (L <= 64) hipLaunchKernelGGL((k_cot_attention<48, 1>), grid, dim3(256), 0, s, qkv, out, L, bs, ts, heads);
  else if (L <= 256) hipLaunchKernelGGL((k_cot_attention<48, 4>), grid, dim3(256), 0, s, qkv, out, L, bs, ts, heads);
  else if (L <= 1024) hipLaunchKernelGGL((k_cot_attention<48, 16>), grid, dim3(256), 0, s, qkv, out, L, bs, ts, heads);
  else return SAMPT_ERR_UNSUPPORTED;
  SAMPT_CHECK_LAUNCH("cot_attention");
  return SAMPT_OK;
}

// end of a window: visibility logit of every (point, frame); carry buffers for the next window; the first S_local frames
// go to the outputs — EVERY active point's, also on frames before its query frame, as upstream's
// `traj_e[:, ind:ind+S, :wind_idx] = coords[-1][:, :S_local]` does.  traj = coords * stride; vis = sigmoid(logit).
__global__ __launch_bounds__(64) void k_cot_window_store(const float* __restrict__ ffeats, const float* __restrict__ vis_w,
                                                          const float* __restrict__ vis_b, const float* __restrict__ coords,
                                                          float stride, int S, int na, int ind, int S_local, int n_total,
                                                          float* __restrict__ coords_prev, float* __restrict__ vis_prev,
                                                          float* __restrict__ traj_out, float* __restrict__ vis_out) {
  const int row = blockIdx.x, pt = row / S, s = row - pt * S, lane = threadIdx.x;
  const float* f = ffeats + (long)row * 128;
  float a = f[lane] * vis_w[lane] + f[lane + 64] * vis_w[lane + 64];
  a = wave_sum(a);
  const float logit = a + vis_b[0];
  if (lane == 0) {
    vis_prev[s * na + pt] = logit;
    if (s < S_local) vis_out[(long)(ind + s) * n_total + pt] = 1.0f / (1.0f + expf(-logit));
  }
  if (lane < 2) {
    const float c = coords[(s * na + pt) * 2 + lane];
    coords_prev[(s * na + pt) * 2 + lane] = c;
    if (s < S_local) traj_out[((long)(ind + s) * n_total + pt) * 2 + lane] = c * stride;
  }
}

int cot_window_store(const float* ffeats, const float* vis_w, const float* vis_b, const float* coords, float stride, int S,
                     int na, int ind, int S_local, int n_total, float* coords_prev, float* vis_prev, float* traj_out,
                     float* vis_out, hipStream_t s) {
  hipLaunchKernelGGL(k_cot_window_store, dim3(na * S), dim3(64), 0, s, ffeats, vis_w, vis_b, coords, stride, S, na, ind,
                     S_local, n_total, coords_prev, vis_prev, traj_out, vis_out);
  SAMPT_CHECK_LAUNCH("cot_window_store");
  return SAMPT_OK;
}

// bilinear resize (F.interpolate, align_corners=False, no antialiasing) of n single-channel planes, uint8 or f32 in, f32 out:
// the adapter's resize of the video to interp_shape (tracker.py:90-92)
template <typename T>
__global__ void k_resize_planes(const T* __restrict__ src, int sh, int sw, float* __restrict__ dst, int dh, int dw,
                                long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % dw);
  const long r = i / dw;
  const int y = (int)(r % dh);
  const long n = r / dh;
  float sy = ((float)sh / (float)dh) * ((float)y + 0.5f) - 0.5f, sx = ((float)sw / (float)dw) * ((float)x + 0.5f) - 0.5f;
  sy = sy < 0.f ? 0.f : sy, sx = sx < 0.f ? 0.f : sx;
  const int y0 = min((int)sy, sh - 1), x0 = min((int)sx, sw - 1);
  const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  const T* b = src + n * sh * sw;
  const float v00 = (float)b[(long)y0 * sw + x0], v01 = (float)b[(long)y0 * sw + x1];
  const float v10 = (float)b[(long)y1 * sw + x0], v11 = (float)b[(long)y1 * sw + x1];
  dst[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

int resize_planes(const void* src, int src_u8, long n, int sh, int sw, float* dst, int dh, int dw, hipStream_t s) {
  const long total = n * dh * dw;
  if (total <= 0) return SAMPT_ERR_ARG;
  if (src_u8)
    hipLaunchKernelGGL(k_resize_planes<uint8_t>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const uint8_t*)src, sh, sw, dst,
                       dh, dw, total);
  else
    hipLaunchKernelGGL(k_resize_planes<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)src, sh, sw, dst, dh,
                       dw, total);
  SAMPT_CHECK_LAUNCH("resize_planes");
  return SAMPT_OK;
}

}  // namespace sampt
