// Prompt encoder / mask decoder helper kernels (all fp32, latency-bound).
#include "ops.h"

namespace sampt {

// ---------------------------------------------------------------------------------------------
// sparse prompt tokens: random-Fourier positional encoding + label embeddings.
// rows: k points, then (no box) one "not a point" pad row | (box) two corner rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_sam_prompt_tokens(const float* __restrict__ pts, const int* __restrict__ labels,
                                                           int k, const float* __restrict__ box,
                                                           const float* __restrict__ gauss,
                                                           const float* __restrict__ point_emb,
                                                           const float* __restrict__ not_a_point, float img_size,
                                                           float* __restrict__ out) {
  const int row = blockIdx.x, j = threadIdx.x;
  float x, y;
  int label;
  const float* emb;
  if (row < k) {
    x = pts[row * 2], y = pts[row * 2 + 1];
    label = labels[row];
    emb = label == 0 ? point_emb : (label == 1 ? point_emb + 256 : not_a_point);
  } else if (box == nullptr) {
    x = 0.f, y = 0.f, label = -1, emb = not_a_point;
  } else {
    int c = row - k;  // corner 0 = (x0,y0), corner 1 = (x1,y1)
    x = box[c * 2], y = box[c * 2 + 1];
    label = 2 + c;
    emb = point_emb + (2 + c) * 256;
  }
  float s = 0.f, c = 0.f;
  if (label != -1) {
    float cx = (x + 0.5f) / img_size, cy = (y + 0.5f) / img_size;
    cx = 2.f * cx - 1.f;
    cy = 2.f * cy - 1.f;
    float v = cx * gauss[j] + cy * gauss[128 + j];
    v = 6.283185307179586f * v;
    s = sinf(v);
    c = cosf(v);
  }
  out[row * 256 + j] = s + emb[j];
  out[row * 256 + 128 + j] = c + emb[128 + j];
}

int sam_prompt_tokens(const float* pts, const int* labels, int k, const float* box, const float* gauss,
                      const float* point_emb, const float* not_a_point, float img_size, float* tokens_out,
                      hipStream_t s) {
  int rows = k + (box ? 2 : 1);
  hipLaunchKernelGGL(k_sam_prompt_tokens, dim3(rows), dim3(128), 0, s, pts, labels, k, box, gauss, point_emb,
                     not_a_point, img_size, tokens_out);
  SAMPT_CHECK_LAUNCH("sam_prompt_tokens");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// attention with many keys and few queries: one workgroup per (query, head); scores stay in registers
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void k_attn_rowblock(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ out, int Nk,
                                                       int ld) {
  constexpr int KPT = 16;  // keys per thread (Nk <= 4096)
  __shared__ float red[8];
  __shared__ float accs[4][HD];
  const int qi = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float qv[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) qv[c] = q[(long)qi * ld + h * HD + c];
  const float inv = sqrtf((float)HD);
  float sc[KPT];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    int key = tid + 256 * i;
    sc[i] = -INFINITY;
    if (key < Nk) {
      const float4* kp = (const float4*)(k + (long)key * ld + h * HD);
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        float4 kk = kp[c];
        a += qv[4 * c] * kk.x + qv[4 * c + 1] * kk.y + qv[4 * c + 2] * kk.z + qv[4 * c + 3] * kk.w;
      }
      sc[i] = a / inv;
      m = fmaxf(m, sc[i]);
    }
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    int key = tid + 256 * i;
    if (key < Nk) {
      float p = expf(sc[i] - m);
      sum += p;
      const float4* vp = (const float4*)(v + (long)key * ld + h * HD);
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        float4 vv = vp[c];
        acc[4 * c] += p * vv.x;
        acc[4 * c + 1] += p * vv.y;
        acc[4 * c + 2] += p * vv.z;
        acc[4 * c + 3] += p * vv.w;
      }
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    float a = wave_sum(acc[c]);
    if (lane == 0) accs[wave][c] = a;
  }
  __syncthreads();
  if (tid < HD) {
    float tot = (red[4] + red[5]) + (red[6] + red[7]);
    float a = (accs[0][tid] + accs[1][tid]) + (accs[2][tid] + accs[3][tid]);
    out[(long)qi * ld + h * HD + tid] = a / tot;
  }
}

int attn_rowblock(const float* q, const float* k, const float* v, float* out, int Nq, int Nk, int heads, int hd,
                  hipStream_t s) {
  if (Nk > 4096 || Nk <= 0 || Nq <= 0) return SAMPT_ERR_ARG;
  int ld = heads * hd;
  if (hd == 16) hipLaunchKernelGGL(k_attn_rowblock<16>, dim3(Nq, heads), dim3(256), 0, s, q, k, v, out, Nk, ld);
  else if (hd == 32) hipLaunchKernelGGL(k_attn_rowblock<32>, dim3(Nq, heads), dim3(256), 0, s, q, k, v, out, Nk, ld);
  else return SAMPT_ERR_UNSUPPORTED;
  SAMPT_CHECK_LAUNCH("attn_rowblock");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// attention with few keys and many queries (image -> token): one thread per (query, head), K/V in LDS
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void k_attn_fewkeys(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, float* __restrict__ out, int Nq,
                                                      int Nk, int heads) {
  extern __shared__ float kv[];  // [2][Nk][ld]
  const int ld = heads * HD;
  float* ks = kv;
  float* vs = kv + (long)Nk * ld;
  for (int i = threadIdx.x; i < Nk * ld; i += 256) {
    ks[i] = k[i];
    vs[i] = v[i];
  }
  __syncthreads();
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)Nq * heads) return;
  int qi = (int)(idx / heads), h = (int)(idx % heads);
  float qv[HD];
  const float4* qp = (const float4*)(q + (long)qi * ld + h * HD);
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    float4 t = qp[c];
    qv[4 * c] = t.x, qv[4 * c + 1] = t.y, qv[4 * c + 2] = t.z, qv[4 * c + 3] = t.w;
  }
  const float inv = sqrtf((float)HD);
  float m = -INFINITY;
  for (int key = 0; key < Nk; ++key) {
    const float* kp = ks + key * ld + h * HD;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) a += qv[c] * kp[c];
    m = fmaxf(m, a / inv);
  }
  float acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = 0.f;
  float sum = 0.f;
  for (int key = 0; key < Nk; ++key) {
    const float* kp = ks + key * ld + h * HD;
    const float* vp = vs + key * ld + h * HD;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) a += qv[c] * kp[c];
    float p = expf(a / inv - m);
    sum += p;
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] += p * vp[c];
  }
  float4* op = (float4*)(out + (long)qi * ld + h * HD);
#pragma unroll
  for (int c = 0; c < HD / 4; ++c)
    op[c] = make_float4(acc[4 * c] / sum, acc[4 * c + 1] / sum, acc[4 * c + 2] / sum, acc[4 * c + 3] / sum);
}

int attn_fewkeys(const float* q, const float* k, const float* v, float* out, int Nq, int Nk, int heads, int hd,
                 hipStream_t s) {
  if (Nk <= 0 || Nk > 64 || hd != 16) return SAMPT_ERR_UNSUPPORTED;
  size_t sh = (size_t)2 * Nk * heads * hd * sizeof(float);
  hipLaunchKernelGGL(k_attn_fewkeys<16>, dim3(cdiv((long)Nq * heads, 256)), dim3(256), sh, s, q, k, v, out, Nq, Nk,
                     heads);
  SAMPT_CHECK_LAUNCH("attn_fewkeys");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// low_res[p] = <hyper, upscaled[p]>   (masks = hyper_in @ upscaled, App. A-4)
// ---------------------------------------------------------------------------------------------
__global__ void k_sam_mask_dot(const float* __restrict__ up, const float* __restrict__ hyper, float* __restrict__ low,
                               int npix, int C) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const float4* u = (const float4*)(up + (long)p * C);
  float a = 0.f;
  for (int c = 0; c < C / 4; ++c) {
    float4 t = u[c];
    a += hyper[4 * c] * t.x + hyper[4 * c + 1] * t.y + hyper[4 * c + 2] * t.z + hyper[4 * c + 3] * t.w;
  }
  low[p] = a;
}

int sam_mask_dot(const float* up, const float* hyper, float* low_res, int npix, int C, hipStream_t s) {
  if (C % 4) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_sam_mask_dot, dim3(cdiv(npix, 256)), dim3(256), 0, s, up, hyper, low_res, npix, C);
  SAMPT_CHECK_LAUNCH("sam_mask_dot");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// Sam.postprocess_masks fused: low (LxL) --bilinear--> (img x img) --crop (in_h,in_w)--> bilinear --> (oh,ow),
// both align_corners=False.  Optionally accumulates the bounding box / count of logits > 0 (sam_pt.py:809-820).
// bbox state: int[5] = {xmin, ymin, xmax, ymax, count}, must be initialised with bbox_state_init.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(int d, float scale, int in, int& i0, int& i1, float& l1) {
  float src = scale * ((float)d + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

__device__ __forceinline__ float up_sample(const float* __restrict__ low, int L, float s1, int Y, int X) {
  int y0, y1, x0, x1;
  float ly, lx;
  src_index(Y, s1, L, y0, y1, ly);
  src_index(X, s1, L, x0, x1, lx);
  float hy = 1.f - ly, hx = 1.f - lx;
  return hy * (hx * low[y0 * L + x0] + lx * low[y0 * L + x1]) + ly * (hx * low[y1 * L + x0] + lx * low[y1 * L + x1]);
}

__global__ __launch_bounds__(256) void k_sam_postprocess(const float* __restrict__ low, int L, int img, int in_h,
                                                         int in_w, float* __restrict__ out, int oh, int ow,
                                                         int* __restrict__ bbox) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  bool pos = false;
  if (x < ow && y < oh) {
    const float s1 = (float)L / (float)img;
    const float sy = (float)in_h / (float)oh, sx = (float)in_w / (float)ow;
    int Y0, Y1, X0, X1;
    float ly, lx;
    src_index(y, sy, in_h, Y0, Y1, ly);
    src_index(x, sx, in_w, X0, X1, lx);
    float v00 = up_sample(low, L, s1, Y0, X0);
    float v;
    if (ly == 0.f && lx == 0.f) {
      v = v00;  // identity second resize (the reference pipelines feed longest-side-1024 frames)
    } else {
      float v01 = up_sample(low, L, s1, Y0, X1), v10 = up_sample(low, L, s1, Y1, X0), v11 = up_sample(low, L, s1, Y1, X1);
      float hy = 1.f - ly, hx = 1.f - lx;
      v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
    out[(long)y * ow + x] = v;
    pos = v > 0.f;
  }
  if (bbox) {
    // deterministic two-stage reduction: per-workgroup partial {xmin,ymin,xmax,ymax,count} -> k_bbox_final
    __shared__ int red[4][5];
    int xmin = pos ? x : 0x7fffffff, xmax = pos ? x : -1, ymin = pos ? y : 0x7fffffff, ymax = pos ? y : -1;
    int cnt = pos ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      xmin = min(xmin, __shfl_xor(xmin, o, 64));
      xmax = max(xmax, __shfl_xor(xmax, o, 64));
      ymin = min(ymin, __shfl_xor(ymin, o, 64));
      ymax = max(ymax, __shfl_xor(ymax, o, 64));
      cnt += __shfl_xor(cnt, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
      red[wave][0] = xmin, red[wave][1] = ymin, red[wave][2] = xmax, red[wave][3] = ymax, red[wave][4] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int* o = bbox + 5 * (blockIdx.y * gridDim.x + blockIdx.x);
      o[0] = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
      o[1] = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
      o[2] = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
      o[3] = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
      o[4] = red[0][4] + red[1][4] + red[2][4] + red[3][4];
    }
  }
}

__global__ __launch_bounds__(256) void k_bbox_final(const int* __restrict__ partial, int nblocks, int* __restrict__ bbox) {
  __shared__ int red[4][5];
  int xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -1, ymax = -1, cnt = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    const int* o = partial + 5 * i;
    xmin = min(xmin, o[0]), ymin = min(ymin, o[1]), xmax = max(xmax, o[2]), ymax = max(ymax, o[3]), cnt += o[4];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    xmin = min(xmin, __shfl_xor(xmin, o, 64));
    xmax = max(xmax, __shfl_xor(xmax, o, 64));
    ymin = min(ymin, __shfl_xor(ymin, o, 64));
    ymax = max(ymax, __shfl_xor(ymax, o, 64));
    cnt += __shfl_xor(cnt, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = xmin, red[wave][1] = ymin, red[wave][2] = xmax, red[wave][3] = ymax, red[wave][4] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bbox[0] = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
    bbox[1] = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
    bbox[2] = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
    bbox[3] = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
    bbox[4] = red[0][4] + red[1][4] + red[2][4] + red[3][4];
  }
}

size_t bbox_partial_ints(int oh, int ow) { return (size_t)5 * cdiv(ow, 64) * cdiv(oh, 4); }

__global__ void k_bbox_state_init(int* bbox) {
  bbox[0] = 0x7fffffff;
  bbox[1] = 0x7fffffff;
  bbox[2] = -1;
  bbox[3] = -1;
  bbox[4] = 0;
}

int bbox_state_init(int* bbox, hipStream_t s) {
  hipLaunchKernelGGL(k_bbox_state_init, dim3(1), dim3(1), 0, s, bbox);
  SAMPT_CHECK_LAUNCH("bbox_state_init");
  return SAMPT_OK;
}

int sam_postprocess_bbox(const float* low, int L, int img, int in_h, int in_w, float* out, int oh, int ow, int* bbox,
                         int* bbox_partial, hipStream_t s) {
  if (in_h > img || in_w > img || (bbox && !bbox_partial)) return SAMPT_ERR_ARG;
  dim3 grid(cdiv(ow, 64), cdiv(oh, 4));
  hipLaunchKernelGGL(k_sam_postprocess, grid, dim3(256), 0, s, low, L, img, in_h, in_w, out, oh, ow,
                     bbox ? bbox_partial : nullptr);
  SAMPT_CHECK_LAUNCH("sam_postprocess");
  if (bbox) {
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, s, bbox_partial, (int)(grid.x * grid.y), bbox);
    SAMPT_CHECK_LAUNCH("bbox_final");
  }
  return SAMPT_OK;
}

int sam_postprocess(const float* low, int L, int img, int in_h, int in_w, float* out, int oh, int ow, hipStream_t s) {
  return sam_postprocess_bbox(low, L, img, in_h, in_w, out, oh, ow, nullptr, nullptr, s);
}

// standalone bbox of logits > 0
__global__ __launch_bounds__(256) void k_bbox_from_logits(const float* __restrict__ logits, int h, int w,
                                                          int* __restrict__ bbox) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  bool pos = x < w && y < h && logits[(long)y * w + x] > 0.f;
  int xmin = pos ? x : 0x7fffffff, xmax = pos ? x : -1, ymin = pos ? y : 0x7fffffff, ymax = pos ? y : -1;
  int cnt = pos ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    xmin = min(xmin, __shfl_xor(xmin, o, 64));
    xmax = max(xmax, __shfl_xor(xmax, o, 64));
    ymin = min(ymin, __shfl_xor(ymin, o, 64));
    ymax = max(ymax, __shfl_xor(ymax, o, 64));
    cnt += __shfl_xor(cnt, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && cnt > 0) {
    atomicMin(&bbox[0], xmin);
    atomicMin(&bbox[1], ymin);
    atomicMax(&bbox[2], xmax);
    atomicMax(&bbox[3], ymax);
    atomicAdd(&bbox[4], cnt);
  }
}

__global__ void k_bbox_to_float(const int* __restrict__ bbox, float* __restrict__ box_out, int* __restrict__ count_out) {
  int i = threadIdx.x;
  if (i < 4) box_out[i] = (float)bbox[i];
  if (i == 4 && count_out) count_out[0] = bbox[4];
}

int bbox_to_float(const int* bbox, float* box_out, int* count_out, hipStream_t s) {
  hipLaunchKernelGGL(k_bbox_to_float, dim3(1), dim3(64), 0, s, bbox, box_out, count_out);
  SAMPT_CHECK_LAUNCH("bbox_to_float");
  return SAMPT_OK;
}

int bbox_from_logits_state(const float* logits, int h, int w, int* bbox_state, hipStream_t s) {
  SAMPT_TRY(bbox_state_init(bbox_state, s));
  hipLaunchKernelGGL(k_bbox_from_logits, dim3(cdiv(w, 64), cdiv(h, 4)), dim3(256), 0, s, logits, h, w, bbox_state);
  SAMPT_CHECK_LAUNCH("bbox_from_logits");
  return SAMPT_OK;
}

}  // namespace sampt

// =============================================================================================
// mask-input embedding, refinement gating, commit and IoU-threshold finalisation
// =============================================================================================
namespace sampt {

// stage A: conv2x2 s2 (1 -> C1) + LayerNorm2d(C1) + GELU ; stage B: conv2x2 s2 (C1 -> C2) + LayerNorm2d + GELU
template <int CIN, int COUT>
__global__ void k_mask_down(const float* __restrict__ in, int ih, int iw, const float* __restrict__ w,
                            const float* __restrict__ b, const float* __restrict__ lnw, const float* __restrict__ lnb,
                            float* __restrict__ out) {
  // in: [ih][iw][CIN] NHWC ; out: [ih/2][iw/2][COUT] ; w: [COUT][CIN][2][2]
  int oh = ih / 2, ow = iw / 2;
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= oh * ow) return;
  int y = p / ow, x = p - y * ow;
  float v[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float a = b[co];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int ky = 0; ky < 2; ++ky)
#pragma unroll
        for (int kx = 0; kx < 2; ++kx)
          a += w[((co * CIN + ci) * 2 + ky) * 2 + kx] * in[((long)(2 * y + ky) * iw + 2 * x + kx) * CIN + ci];
    v[co] = a;
  }
  float mean = 0.f;
#pragma unroll
  for (int co = 0; co < COUT; ++co) mean += v[co];
  mean /= (float)COUT;
  float var = 0.f;
#pragma unroll
  for (int co = 0; co < COUT; ++co) var += (v[co] - mean) * (v[co] - mean);
  var /= (float)COUT;
  float rstd = 1.0f / sqrtf(var + 1e-6f);
#pragma unroll
  for (int co = 0; co < COUT; ++co) out[(long)p * COUT + co] = gelu_erf((v[co] - mean) * rstd * lnw[co] + lnb[co]);
}

// stage C: src[p][c] = feat[p][c] + b2[c] + sum_k w2[c][k] * e[p][k]
__global__ void k_mask_embed_out(const float* __restrict__ e, int C2, const float* __restrict__ w2,
                                 const float* __restrict__ b2, const float* __restrict__ feat, float* __restrict__ src,
                                 int npix) {
  int p = blockIdx.x, c = threadIdx.x;  // 256 threads = output channels
  if (p >= npix) return;
  float a = b2[c];
  for (int k = 0; k < C2; ++k) a += w2[c * C2 + k] * e[(long)p * C2 + k];
  src[(long)p * 256 + c] = feat[(long)p * 256 + c] + a;
}

int sam_mask_embed_src(const float* mask, int g, const MaskEmbedW& w, const float* feat, float* tmp0, float* tmp1,
                       float* src, hipStream_t s) {
  int L = 4 * g;
  hipLaunchKernelGGL((k_mask_down<1, 4>), dim3(cdiv((L / 2) * (L / 2), 256)), dim3(256), 0, s, mask, L, L, w.w0, w.b0,
                     w.ln0w, w.ln0b, tmp0);
  SAMPT_CHECK_LAUNCH("mask_down0");
  hipLaunchKernelGGL((k_mask_down<4, 16>), dim3(cdiv(g * g, 256)), dim3(256), 0, s, tmp0, L / 2, L / 2, w.w1, w.b1,
                     w.ln1w, w.ln1b, tmp1);
  SAMPT_CHECK_LAUNCH("mask_down1");
  hipLaunchKernelGGL(k_mask_embed_out, dim3(g * g), dim3(256), 0, s, tmp1, 16, w.w2, w.b2, feat, src, g * g);
  SAMPT_CHECK_LAUNCH("mask_embed_out");
  return SAMPT_OK;
}

__global__ void k_sam_refine_gate(int* active, const int* __restrict__ bbox_cur, float* __restrict__ box_f) {
  int i = threadIdx.x;
  if (i < 4) box_f[i] = (float)bbox_cur[i];
  if (i == 0) active[0] = (active[0] != 0 && bbox_cur[4] >= 2) ? 1 : 0;
}

int sam_refine_gate(int* active, const int* bbox_cur, float* box_f, hipStream_t s) {
  hipLaunchKernelGGL(k_sam_refine_gate, dim3(1), dim3(64), 0, s, active, bbox_cur, box_f);
  SAMPT_CHECK_LAUNCH("sam_refine_gate");
  return SAMPT_OK;
}

__global__ void k_sam_commit(const int* __restrict__ active, const float* __restrict__ cand_logits,
                             float* __restrict__ cur_logits, long n_logits, const float* __restrict__ cand_low,
                             float* __restrict__ cur_low, long n_low, const float* __restrict__ cand_iou,
                             float* __restrict__ cur_iou, const int* __restrict__ cand_bbox, int* __restrict__ cur_bbox) {
  if (active[0] == 0) return;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_logits) cur_logits[i] = cand_logits[i];
  if (i < n_low) cur_low[i] = cand_low[i];
  if (i == 0) cur_iou[0] = cand_iou[0];
  if (i < 5) cur_bbox[i] = cand_bbox[i];
}

int sam_commit(const int* active, const float* cand_logits, float* cur_logits, long n_logits, const float* cand_low,
               float* cur_low, long n_low, const float* cand_iou, float* cur_iou, const int* cand_bbox, int* cur_bbox,
               hipStream_t s) {
  long n = n_logits > n_low ? n_logits : n_low;
  hipLaunchKernelGGL(k_sam_commit, dim3(cdiv(n, 256)), dim3(256), 0, s, active, cand_logits, cur_logits, n_logits,
                     cand_low, cur_low, n_low, cand_iou, cur_iou, cand_bbox, cur_bbox);
  SAMPT_CHECK_LAUNCH("sam_commit");
  return SAMPT_OK;
}

__global__ void k_sam_finalize_mask(const float* __restrict__ logits, const float* __restrict__ iou, float thr,
                                    float* __restrict__ out, float* __restrict__ score_out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float sc = iou[0];
  if (i < n) out[i] = sc < thr ? -INFINITY : logits[i];
  if (i == 0) score_out[0] = sc;
}

int sam_finalize_mask(const float* logits, const float* iou, float thr, float* out, float* score_out, long n,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_sam_finalize_mask, dim3(cdiv(n, 256)), dim3(256), 0, s, logits, iou, thr, out, score_out, n);
  SAMPT_CHECK_LAUNCH("sam_finalize_mask");
  return SAMPT_OK;
}

}  // namespace sampt
