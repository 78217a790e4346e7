// HBM-bound elementwise / normalisation / resampling kernels (NHWC, f32 unless noted).
// All of them are pure streaming kernels: one pass over the data, 16-byte accesses where the layout allows,
// grids of >> 256 workgroups so that all 8 XCDs stay busy.
#include "ops.h"

namespace sampt {

// ---------------------------------------------------------------------------------------------
// uint8 CHW -> normalised f32 NHWC4
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_rgb_u8chw_to_nhwc4(const T* __restrict__ src, float4* __restrict__ dst, long npix_total, long hw) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix_total) return;
  long t = i / hw, p = i - t * hw;
  const T* b = src + t * 3 * hw + p;
  float r = 2.0f * ((float)b[0] / 255.0f) - 1.0f;
  float g = 2.0f * ((float)b[hw] / 255.0f) - 1.0f;
  float bl = 2.0f * ((float)b[2 * hw] / 255.0f) - 1.0f;
  dst[i] = make_float4(r, g, bl, 0.f);
}

// src: uint8 frames, or (src_f32 != 0) float frames with values in [0, 255] (a resized video, PIPS++ image_size)
int rgb_u8chw_to_nhwc4(const void* src, int src_f32, float* dst, int T, int H, int W, hipStream_t s) {
  long hw = (long)H * W, n = hw * T;
  if (src_f32)
    hipLaunchKernelGGL(k_rgb_u8chw_to_nhwc4<float>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const float*)src, (float4*)dst, n,
                       hw);
  else
    hipLaunchKernelGGL(k_rgb_u8chw_to_nhwc4<uint8_t>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const uint8_t*)src,
                       (float4*)dst, n, hw);
  SAMPT_CHECK_LAUNCH("rgb_u8chw_to_nhwc4");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// Row gather / scatter by an index vector of (frame, object) ITEM numbers, item = t * nm + m (SamPt's decode staging: a replayed
// hipGraph reads and writes fixed buffers, so a chunk's embeddings are gathered into them and its masks / scores scattered out):
//   gather : dst[r] = src[idx[r] / nm]                       (the embedding of item r's frame)
//   scatter: dst[(idx[r] % nm) * nf + idx[r] / nm] = src[r]   (logits [nm][nf][H*W]; scores: nm = 1 -> dst[idx[r]])
// rows of row16 16-byte units; one workgroup walks a row segment of 256 units
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_move_rows(const float4* __restrict__ src, float4* __restrict__ dst, const int* __restrict__ idx,
                                                   long row16, int nm, int nf, int scatter) {
  const long r = blockIdx.y, u = (long)blockIdx.x * 256 + threadIdx.x;
  if (u >= row16) return;
  const int it = idx[r];
  const long sr = scatter ? r : it / nm, dr = scatter ? (long)(it % nm) * nf + it / nm : r;
  dst[dr * row16 + u] = src[sr * row16 + u];
}

__global__ void k_move_words(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ idx, int rows, int nm,
                             int nf, int scatter) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int it = idx[r];
  if (scatter) dst[(long)(it % nm) * nf + it / nm] = src[r];
  else dst[r] = src[it / nm];
}

static int move_words(const void* src, void* dst, const int* idx, int rows, int nm, int nf, int scatter, hipStream_t s) {
  hipLaunchKernelGGL(k_move_words, dim3(cdiv(rows, 256)), dim3(256), 0, s, (const float*)src, (float*)dst, idx, rows, nm, nf, scatter);
  SAMPT_CHECK_LAUNCH("move_words");
  return SAMPT_OK;
}

int move_rows(const void* src, void* dst, const int* idx, int rows, long row_bytes, int nm, int nf, int scatter, hipStream_t s) {
  if (!src || !dst || !idx || rows <= 0 || row_bytes <= 0 || nm <= 0) return SAMPT_ERR_ARG;
  if ((row_bytes & 15) || (((uintptr_t)src | (uintptr_t)dst) & 15)) {      // short rows (scores: 4 bytes): one thread per row
    if (row_bytes != 4) return SAMPT_ERR_UNSUPPORTED;
    return move_words(src, dst, idx, rows, nm, nf, scatter, s);
  }
  const long row16 = row_bytes / 16;
  hipLaunchKernelGGL(k_move_rows, dim3((unsigned)cdiv(row16, 256), rows), dim3(256), 0, s, (const float4*)src, (float4*)dst, idx, row16,
                     nm, nf, scatter);
  SAMPT_CHECK_LAUNCH("move_rows");
  return SAMPT_OK;
}

__global__ void k_fill_f32(float* __restrict__ dst, long n, float v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}

int fill_f32(float* dst, long n, float v, hipStream_t s) {
  if (!dst || n <= 0) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, dst, n, v);
  SAMPT_CHECK_LAUNCH("fill_f32");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// InstanceNorm2d statistics: deterministic two-stage reduction in fp64
// ---------------------------------------------------------------------------------------------
static constexpr int IN_PIX_PER_BLOCK = 512;

size_t instnorm_partial_doubles(int nimg, long hw, int C) {
  return (size_t)nimg * cdiv(hw, IN_PIX_PER_BLOCK) * C * 2;
}

__global__ void k_instnorm_partial(const float* __restrict__ x, long hw, int C, double* __restrict__ part) {
  // block = (C, ny); grid = (nchunks, nimg)
  extern __shared__ double sh[];  // [ny][C][2]
  const int c = threadIdx.x, ty = threadIdx.y, ny = blockDim.y;
  const long img = blockIdx.y, chunk = blockIdx.x;
  long p0 = chunk * IN_PIX_PER_BLOCK, p1 = p0 + IN_PIX_PER_BLOCK;
  if (p1 > hw) p1 = hw;
  const float* base = x + img * hw * C + c;
  double s1 = 0.0, s2 = 0.0;
  for (long p = p0 + ty; p < p1; p += ny) {
    double v = (double)base[p * C];
    s1 += v;
    s2 += v * v;
  }
  sh[(ty * C + c) * 2] = s1;
  sh[(ty * C + c) * 2 + 1] = s2;
  __syncthreads();
  if (ty == 0) {
    for (int j = 1; j < ny; ++j) {
      s1 += sh[(j * C + c) * 2];
      s2 += sh[(j * C + c) * 2 + 1];
    }
    double* o = part + ((img * gridDim.x + chunk) * C + c) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

// second stage: the chunk partials of one image, summed in fp64 in a fixed order (thread (c, ty) takes chunks ty, ty + NY, ...;
// the NY partial sums are then added in ty order); workgroup (img, channel group of blockDim.x channels)
__global__ void k_instnorm_final(const double* __restrict__ part, int nchunks, long hw, int C, float eps,
                                 float* __restrict__ mean_rstd) {
  extern __shared__ double shf[];   // [NY][CB][2]
  const int CB = blockDim.x, cl = threadIdx.x, c = blockIdx.y * CB + cl, ty = threadIdx.y, NY = blockDim.y;
  const long img = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int j = ty; j < nchunks; j += NY) {
      const double* o = part + ((img * nchunks + j) * C + c) * 2;
      s1 += o[0];
      s2 += o[1];
    }
  shf[(ty * CB + cl) * 2] = s1, shf[(ty * CB + cl) * 2 + 1] = s2;
  __syncthreads();
  if (ty != 0 || c >= C) return;
  for (int t = 1; t < NY; ++t) s1 += shf[(t * CB + cl) * 2], s2 += shf[(t * CB + cl) * 2 + 1];
  double mean = s1 / (double)hw;
  double var = s2 / (double)hw - mean * mean;
  if (var < 0.0) var = 0.0;
  mean_rstd[(img * C + c) * 2] = (float)mean;
  mean_rstd[(img * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// (channel groups of 32: 8 images x 64 channels used to be 8 workgroups walking 288 - 576 chunks each — 5 - 21 us of latency chain)
static void launch_instnorm_final(const double* partials, int nimg, int nchunks, long hw, int C, float eps, float* mean_rstd,
                                  hipStream_t s) {
  const int CB = C < 32 ? C : 32, NY = 1024 / CB > 32 ? 32 : 1024 / CB;
  hipLaunchKernelGGL(k_instnorm_final, dim3(nimg, cdiv(C, CB)), dim3(CB, NY), (size_t)NY * CB * 2 * sizeof(double), s, partials, nchunks,
                     hw, C, eps, mean_rstd);
}

// C % 4 == 0: a thread owns 4 channels (16-byte loads, two pixels in flight per iteration) — the scalar kernel above
// ran at half the bandwidth of k_instnorm_apply although it only reads.  Same partial layout, sums in fp64.
__global__ __launch_bounds__(256) void k_instnorm_partial_v4(const float* __restrict__ x, long hw, int C, int ny,
                                                             double* __restrict__ part) {
  extern __shared__ double sh[];  // [ny][C][2]
  const int c4n = C >> 2, tid = threadIdx.x;
  const int cq = tid % c4n, ty = tid / c4n;
  const long img = blockIdx.y, chunk = blockIdx.x;
  long p0 = chunk * IN_PIX_PER_BLOCK, p1 = p0 + IN_PIX_PER_BLOCK;
  if (p1 > hw) p1 = hw;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (ty < ny) {
    const float* base = x + img * hw * C + cq * 4;
    long p = p0 + ty;
    for (; p + ny < p1; p += 2 * ny) {
      const float4 a = *(const float4*)(base + p * C), b = *(const float4*)(base + (p + ny) * C);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double u = (double)av[j], w = (double)bv[j];
        s1[j] += u, s2[j] += u * u;
        s1[j] += w, s2[j] += w * w;
      }
    }
    if (p < p1) {
      const float4 a = *(const float4*)(base + p * C);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double u = (double)av[j];
        s1[j] += u, s2[j] += u * u;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sh[(ty * C + cq * 4 + j) * 2] = s1[j];
      sh[(ty * C + cq * 4 + j) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  if (tid < C) {
    double t1 = 0.0, t2 = 0.0;
    for (int j = 0; j < ny; ++j) {
      t1 += sh[(j * C + tid) * 2];
      t2 += sh[(j * C + tid) * 2 + 1];
    }
    double* o = part + ((img * gridDim.x + chunk) * C + tid) * 2;
    o[0] = t1;
    o[1] = t2;
  }
}

// second stage alone: the partials were written by the producer of the map (conv3x3_halo_x3 with GemmP::in_part)
int instnorm_finalize(const double* partials, int nimg, int nchunks, long hw, int C, float eps, float* mean_rstd, hipStream_t s) {
  if (C > 1024 || C <= 0 || nchunks <= 0) return SAMPT_ERR_ARG;
  launch_instnorm_final(partials, nimg, nchunks, hw, C, eps, mean_rstd, s);
  SAMPT_CHECK_LAUNCH("instnorm_final");
  return SAMPT_OK;
}

int instnorm_stats(const float* x, int nimg, long hw, int C, float eps, double* partials, float* mean_rstd,
                   hipStream_t s) {
  if (C > 1024 || C <= 0) return SAMPT_ERR_ARG;
  if (C % 4 == 0 && C <= 256 && !((uintptr_t)x & 15)) {
    const int nyv = 256 / (C / 4) > 16 ? 16 : 256 / (C / 4);
    const int nch = cdiv(hw, IN_PIX_PER_BLOCK);
    hipLaunchKernelGGL(k_instnorm_partial_v4, dim3(nch, nimg), dim3(256), (size_t)nyv * C * 2 * sizeof(double), s, x, hw, C,
                       nyv, partials);
    SAMPT_CHECK_LAUNCH("instnorm_partial_v4");
    launch_instnorm_final(partials, nimg, nch, hw, C, eps, mean_rstd, s);
    SAMPT_CHECK_LAUNCH("instnorm_final");
    return SAMPT_OK;
  }
  int ny = 256 / C;
  if (ny < 1) ny = 1;
  int nchunks = cdiv(hw, IN_PIX_PER_BLOCK);
  hipLaunchKernelGGL(k_instnorm_partial, dim3(nchunks, nimg), dim3(C, ny), (size_t)ny * C * 2 * sizeof(double), s, x,
                     hw, C, partials);
  SAMPT_CHECK_LAUNCH("instnorm_partial");
  launch_instnorm_final(partials, nimg, nchunks, hw, C, eps, mean_rstd, s);
  SAMPT_CHECK_LAUNCH("instnorm_final");
  return SAMPT_OK;
}

// HL: additionally write y split into two fp16 planes (hi = fp16(y), lo = fp16(y - hi)) for the split-fp16 convolution that
// consumes it (conv_f16x3.hip, AHL): the split is done once here instead of once per filter tap there
template <bool HL>
__global__ void k_instnorm_apply(const float4* __restrict__ x, const float* __restrict__ mr,
                                 const float4* __restrict__ skip, float4* __restrict__ y, long n4, long hwc4, int c4n,
                                 int relu1, h4* __restrict__ y_hi, h4* __restrict__ y_lo) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  long img = i / hwc4;
  int c = (int)(i % c4n) * 4;
  const float* m = mr + (img * c4n * 4 + c) * 2;
  float4 v = x[i];
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[j] = (o[j] - m[2 * j]) * m[2 * j + 1];
    if (relu1) o[j] = fmaxf(o[j], 0.f);
  }
  if (skip) {
    float4 k = skip[i];
    o[0] = fmaxf(o[0] + k.x, 0.f);
    o[1] = fmaxf(o[1] + k.y, 0.f);
    o[2] = fmaxf(o[2] + k.z, 0.f);
    o[3] = fmaxf(o[3] + k.w, 0.f);
  }
  if (y) y[i] = make_float4(o[0], o[1], o[2], o[3]);        // (null: the map is only ever read as fp16 planes — 4 of 12 bytes less)
  if (HL) {
    const h4 hi = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
    const h4 lo = {(half_t)(o[0] - (float)hi[0]), (half_t)(o[1] - (float)hi[1]), (half_t)(o[2] - (float)hi[2]),
                   (half_t)(o[3] - (float)hi[3])};
    y_hi[i] = hi, y_lo[i] = lo;
  }
}

int instnorm_apply(const float* x, const float* mean_rstd, const float* skip, float* y, int nimg, long hw, int C,
                   int relu1, hipStream_t s, half_t* y_hi, half_t* y_lo) {
  if (C % 4 || (!y && !(y_hi && y_lo))) return SAMPT_ERR_ARG;
  long n4 = (long)nimg * hw * C / 4;
  if (y_hi && y_lo)
    hipLaunchKernelGGL(k_instnorm_apply<true>, dim3(cdiv(n4, 256)), dim3(256), 0, s, (const float4*)x, mean_rstd,
                       (const float4*)skip, (float4*)y, n4, hw * C / 4, C / 4, relu1, (h4*)y_hi, (h4*)y_lo);
  else
    hipLaunchKernelGGL(k_instnorm_apply<false>, dim3(cdiv(n4, 256)), dim3(256), 0, s, (const float4*)x, mean_rstd,
                       (const float4*)skip, (float4*)y, n4, hw * C / 4, C / 4, relu1, (h4*)nullptr, (h4*)nullptr);
  SAMPT_CHECK_LAUNCH("instnorm_apply");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// bilinear resize (torch F.interpolate semantics), NHWC, writes a channel slice of the destination
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilinear_src(int d, int in, int out, int align, int& i0, int& i1, float& l1) {
  float src;
  if (align) {
    float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    src = scale * (float)d;
  } else {
    float scale = (float)in / (float)out;
    src = scale * ((float)d + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

// HL: the result goes out as two fp16 planes (hi = fp16(v), lo = fp16(v - hi); same [n][dh][dw][dstC] layout) instead of f32 —
// the operand format of the split-fp16 convolution that consumes the concatenated map (conv_f16x3.hip, LDS-DMA kernel)
template <bool HL>
__global__ void k_resize_bilinear_nhwc(const float4* __restrict__ src, int sh, int sw, int c4n, float4* __restrict__ dst,
                                       int dh, int dw, int dst_c4n, int c_off4, int align, long total,
                                       h4* __restrict__ dst_hi, h4* __restrict__ dst_lo) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = (int)(i % c4n);
  long r = i / c4n;
  int x = (int)(r % dw);
  r /= dw;
  int y = (int)(r % dh);
  long n = r / dh;
  int y0, y1, x0, x1;
  float ly, lx;
  bilinear_src(y, sh, dh, align, y0, y1, ly);
  bilinear_src(x, sw, dw, align, x0, x1, lx);
  const float4* b = src + n * sh * sw * c4n + c;
  float4 v00 = b[((long)y0 * sw + x0) * c4n], v01 = b[((long)y0 * sw + x1) * c4n];
  float4 v10 = b[((long)y1 * sw + x0) * c4n], v11 = b[((long)y1 * sw + x1) * c4n];
  float hy = 1.f - ly, hx = 1.f - lx;
  float4 o;
  o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
  o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
  o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
  o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
  const long di = ((n * dh + y) * dw + x) * dst_c4n + c_off4 + c;
  if (HL) {
    const h4 hi = {(half_t)o.x, (half_t)o.y, (half_t)o.z, (half_t)o.w};
    const h4 lo = {(half_t)(o.x - (float)hi[0]), (half_t)(o.y - (float)hi[1]), (half_t)(o.z - (float)hi[2]),
                   (half_t)(o.w - (float)hi[3])};
    dst_hi[di] = hi, dst_lo[di] = lo;
  } else {
    dst[di] = o;
  }
}

int resize_bilinear_nhwc(const float* src, int n, int sh, int sw, int C, float* dst, int dh, int dw, int dstC,
                         int c_off, int align_corners, hipStream_t s, half_t* dst_hi, half_t* dst_lo) {
  if (C % 4 || dstC % 4 || c_off % 4) return SAMPT_ERR_ARG;
  long total = (long)n * dh * dw * (C / 4);
  if (dst_hi && dst_lo)
    hipLaunchKernelGGL(k_resize_bilinear_nhwc<true>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float4*)src, sh, sw, C / 4,
                       (float4*)nullptr, dh, dw, dstC / 4, c_off / 4, align_corners, total, (h4*)dst_hi, (h4*)dst_lo);
  else
    hipLaunchKernelGGL(k_resize_bilinear_nhwc<false>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float4*)src, sh, sw, C / 4,
                       (float4*)dst, dh, dw, dstC / 4, c_off / 4, align_corners, total, (h4*)nullptr, (h4*)nullptr);
  SAMPT_CHECK_LAUNCH("resize_bilinear_nhwc");
  return SAMPT_OK;
}

__global__ void k_avgpool2x2(const float4* __restrict__ src, int h, int w, int c4n, float4* __restrict__ dst, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int oh = h / 2, ow = w / 2;
  int c = (int)(i % c4n);
  long r = i / c4n;
  int x = (int)(r % ow);
  r /= ow;
  int y = (int)(r % oh);
  long n = r / oh;
  const float4* b = src + ((n * h + 2 * y) * w + 2 * x) * c4n + c;
  float4 a0 = b[0], a1 = b[c4n], a2 = b[(long)w * c4n], a3 = b[(long)w * c4n + c4n];
  float4 o;
  o.x = (((a0.x + a1.x) + a2.x) + a3.x) / 4.0f;
  o.y = (((a0.y + a1.y) + a2.y) + a3.y) / 4.0f;
  o.z = (((a0.z + a1.z) + a2.z) + a3.z) / 4.0f;
  o.w = (((a0.w + a1.w) + a2.w) + a3.w) / 4.0f;
  dst[i] = o;
}

int avgpool2x2_nhwc(const float* src, int n, int h, int w, int C, float* dst, hipStream_t s) {
  if (C % 4) return SAMPT_ERR_ARG;
  long total = (long)n * (h / 2) * (w / 2) * (C / 4);
  hipLaunchKernelGGL(k_avgpool2x2, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float4*)src, h, w, C / 4,
                     (float4*)dst, total);
  SAMPT_CHECK_LAUNCH("avgpool2x2");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over rows: one wave per row, the row lives in registers (two-pass mean / variance)
// ---------------------------------------------------------------------------------------------
template <int NI>
__global__ __launch_bounds__(256) void k_layernorm_rows(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, void* __restrict__ y, long M, int D,
                                                        float eps, const int* __restrict__ src_rows, int out_f16,
                                                        int act) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  long srow = src_rows ? (long)src_rows[row] : row;
  if (srow < 0) {  // padded row of a window partition: zeros (App. A-3: pad AFTER norm1)
    for (int i = 0; i < NI; ++i) {
      int idx = lane + 64 * i;
      if (idx < D) {
        if (out_f16 == 2) ((half_t*)y)[row * 2 * D + x3_col(idx)] = ((half_t*)y)[row * 2 * D + x3_col(idx) + 32] = (half_t)0.f;
        else if (out_f16) ((half_t*)y)[row * D + idx] = (half_t)0.f;
        else ((float*)y)[row * D + idx] = 0.f;
      }
    }
    return;
  }
  const float* xr = x + srow * D;
  float v[NI];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int idx = lane + 64 * i;
    v[i] = idx < D ? xr[idx] : 0.f;
    sum += v[i];
  }
  float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int idx = lane + 64 * i;
    float d = idx < D ? v[i] - mean : 0.f;
    sq += d * d;
  }
  float var = wave_sum(sq) / (float)D;
  float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int idx = lane + 64 * i;
    if (idx < D) {
      float o = (v[i] - mean) * rstd * w[idx] + b[idx];
      o = apply_act(o, act);
      if (out_f16 == 2) {
        half_t hi, lo;
        split_f16(o, hi, lo);
        half_t* yp = (half_t*)y + row * 2 * D + x3_col(idx);
        yp[0] = hi, yp[32] = lo;
      } else if (out_f16) ((half_t*)y)[row * D + idx] = (half_t)o;
      else ((float*)y)[row * D + idx] = o;
    }
  }
}

// vectorised variant for D % 256 == 0: every lane owns NV float4 chunks (16-byte loads, 8/16-byte stores)
template <int NV>
__global__ __launch_bounds__(256) void k_layernorm_rows_v4(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, void* __restrict__ y, long M,
                                                           int D, float eps, const int* __restrict__ src_rows,
                                                           int out_f16, int act) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  long srow = src_rows ? (long)src_rows[row] : row;
  if (srow < 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = (lane + 64 * i) * 4;
      if (out_f16 == 2) {
        *(h4*)((half_t*)y + row * 2 * D + x3_col(c)) = (h4){0, 0, 0, 0};
        *(h4*)((half_t*)y + row * 2 * D + x3_col(c) + 32) = (h4){0, 0, 0, 0};
      } else if (out_f16 == 3) {
        *(h4*)((half_t*)y + row * D + c) = (h4){0, 0, 0, 0};
        *(h4*)((half_t*)y + (M + row) * D + c) = (h4){0, 0, 0, 0};
      } else if (out_f16) *(h4*)((half_t*)y + row * D + c) = (h4){0, 0, 0, 0};
      else *(float4*)((float*)y + row * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const float4* xr = (const float4*)(x + srow * D);
  float4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = xr[lane + 64 * i];
    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
    // (explicit fused multiply-adds here and below: gemm_x3_wres.hip k_gemm_x3_wres_ln repeats this arithmetic operation for operation)
    sq += __builtin_fmaf(d1, d1, d0 * d0) + __builtin_fmaf(d3, d3, d2 * d2);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    const float4 wv = *(const float4*)(w + c), bv = *(const float4*)(b + c);
    float o0 = apply_act(__builtin_fmaf((v[i].x - mean) * rstd, wv.x, bv.x), act);
    float o1 = apply_act(__builtin_fmaf((v[i].y - mean) * rstd, wv.y, bv.y), act);
    float o2 = apply_act(__builtin_fmaf((v[i].z - mean) * rstd, wv.z, bv.z), act);
    float o3 = apply_act(__builtin_fmaf((v[i].w - mean) * rstd, wv.w, bv.w), act);
    if (out_f16 == 2) {
      h4 hi, lo;
      half_t a, bb;
      split_f16(o0, a, bb), hi[0] = a, lo[0] = bb;
      split_f16(o1, a, bb), hi[1] = a, lo[1] = bb;
      split_f16(o2, a, bb), hi[2] = a, lo[2] = bb;
      split_f16(o3, a, bb), hi[3] = a, lo[3] = bb;
      half_t* yp = (half_t*)y + row * 2 * D + x3_col(c);
      *(h4*)yp = hi;
      *(h4*)(yp + 32) = lo;
    } else if (out_f16 == 3) {      // two fp16 planes [2][M][D] (hi, lo): the pre-split operand of the halo convolution (conv_halo_x3.hip)
      h4 hi, lo;
      half_t a, bb;
      split_f16(o0, a, bb), hi[0] = a, lo[0] = bb;
      split_f16(o1, a, bb), hi[1] = a, lo[1] = bb;
      split_f16(o2, a, bb), hi[2] = a, lo[2] = bb;
      split_f16(o3, a, bb), hi[3] = a, lo[3] = bb;
      half_t* yp = (half_t*)y + row * D + c;
      *(h4*)yp = hi;
      *(h4*)(yp + M * D) = lo;
    } else if (out_f16) *(h4*)((half_t*)y + row * D + c) = (h4){(half_t)o0, (half_t)o1, (half_t)o2, (half_t)o3};
    else *(float4*)((float*)y + row * D + c) = make_float4(o0, o1, o2, o3);
  }
}

// D == 64 (LayerNorm2d between the decoder's two transposed convolutions, 1.6 M pixel rows per pass): 16 lanes per row, one
// float4 each, 16 rows per workgroup; the row sums stay inside a DPP row (quad_perm x2, row_half_mirror, row_mirror)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}

__global__ __launch_bounds__(256) void k_layernorm_rows_d64(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y, long M,
                                                            float eps, int act) {
  const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = (threadIdx.x & 15) * 4;
  const bool live = row < M;
  const float4 v = *(const float4*)(x + (live ? row : 0) * 64 + c);
  const float mean = row16_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 64.0f);
  const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  // (explicit fused multiply-adds: gemm_x3_wres.hip's fused epilogue repeats this arithmetic operation for operation, so a row is the
  //  same bits whether its LayerNorm ran here or there)
  const float rstd = 1.0f / sqrtf(row16_sum(__builtin_fmaf(d1, d1, d0 * d0) + __builtin_fmaf(d3, d3, d2 * d2)) * (1.0f / 64.0f) + eps);
  if (!live) return;
  const float4 wv = *(const float4*)(w + c), bv = *(const float4*)(b + c);
  *(float4*)(y + row * 64 + c) = make_float4(apply_act(__builtin_fmaf(d0 * rstd, wv.x, bv.x), act), apply_act(__builtin_fmaf(d1 * rstd, wv.y, bv.y), act),
                                             apply_act(__builtin_fmaf(d2 * rstd, wv.z, bv.z), act), apply_act(__builtin_fmaf(d3 * rstd, wv.w, bv.w), act));
}

// out[k] = mean over rows r < M of A[rowmap ? rowmap[r] : r][k] for an fp16 matrix: one workgroup per 64 columns, 4 row groups whose
// partial sums meet in LDS in a fixed order (deterministic).  Calibration only (VitEngine::calib): not on the timed path.
__global__ __launch_bounds__(256) void k_colmean_rows_f16(const half_t* __restrict__ A, long M, int K, int lda,
                                                          const int* __restrict__ rowmap, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < K)
    for (long r = rg; r < M; r += 4) {
      const long src = rowmap ? rowmap[r] : r;
      acc += (float)A[src * lda + c];
    }
  part[rg][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rg == 0 && c < K) out[c] = (((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x]) / (float)M;
}

int colmean_rows_f16(const half_t* A, long M, int K, int lda, const int* rowmap, float* out, hipStream_t s) {
  if (!A || !out || M <= 0 || K <= 0) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_colmean_rows_f16, dim3(cdiv(K, 64)), dim3(256), 0, s, A, M, K, lda, rowmap, out);
  SAMPT_CHECK_LAUNCH("colmean_rows_f16");
  return SAMPT_OK;
}

int layernorm_rows(const float* x, const float* w, const float* b, void* y, long M, int D, float eps,
                   const int* src_rows, int out_f16, int act, hipStream_t s) {
  if (D <= 0 || D > 2048 || M <= 0) return SAMPT_ERR_ARG;
  if (out_f16 == 2 && (D % 32)) return SAMPT_ERR_ARG;     // x3 rows are made of whole 32-blocks
  // out_f16 == 3 (two fp16 planes [2][M][D]): the vectorised kernel only
  if (out_f16 == 3 && !(D % 256 == 0 && D <= 1536 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0)))
    return SAMPT_ERR_UNSUPPORTED;
  if (D == 64 && !src_rows && !out_f16 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0)) {
    hipLaunchKernelGGL(k_layernorm_rows_d64, dim3((unsigned)cdiv(M, 16)), dim3(256), 0, s, x, w, b, (float*)y, M, eps, act);
    SAMPT_CHECK_LAUNCH("layernorm_rows_d64");
    return SAMPT_OK;
  }
  dim3 grid(cdiv(M, 4)), block(256);
  if (D % 256 == 0 && D <= 1536 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0)) {
#define LNV(NVv) hipLaunchKernelGGL(k_layernorm_rows_v4<NVv>, grid, block, 0, s, x, w, b, y, M, D, eps, src_rows, out_f16, act)
    switch (D / 256) {
      case 1: LNV(1); break;
      case 2: LNV(2); break;
      case 3: LNV(3); break;
      case 4: LNV(4); break;
      case 5: LNV(5); break;
      default: LNV(6); break;
    }
#undef LNV
    SAMPT_CHECK_LAUNCH("layernorm_rows_v4");
    return SAMPT_OK;
  }
  int ni = cdiv(D, 64);
#define LN(NIv) hipLaunchKernelGGL(k_layernorm_rows<NIv>, grid, block, 0, s, x, w, b, y, M, D, eps, src_rows, out_f16, act)
  if (ni <= 1) LN(1);
  else if (ni <= 4) LN(4);
  else if (ni <= 8) LN(8);
  else if (ni <= 12) LN(12);
  else if (ni <= 16) LN(16);
  else if (ni <= 20) LN(20);
  else LN(32);
#undef LN
  SAMPT_CHECK_LAUNCH("layernorm_rows");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void k_add_bcast(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n,
                            long bmod) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = a[i] + b[i % bmod];
}

int add_bcast(const float* a, const float* b, float* out, long n, long bmod, hipStream_t s) {
  if (n <= 0 || bmod <= 0) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_add_bcast, dim3(cdiv(n, 256)), dim3(256), 0, s, a, b, out, n, bmod);
  SAMPT_CHECK_LAUNCH("add_bcast");
  return SAMPT_OK;
}

__global__ void k_cast_f32_f16(const float* __restrict__ x, half_t* __restrict__ y, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (half_t)x[i];
}

// f32 rows [M][K] -> x3 rows [M][2K] (common.h): one thread per 4 consecutive columns
__global__ void k_split_rows_x3(const float4* __restrict__ x, half_t* __restrict__ y, long n4, int K4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const long row = i / K4;
  const int c = (int)(i - row * K4) * 4;
  const float4 v = x[i];
  h4 hi, lo;
  half_t a, b;
  split_f16(v.x, a, b), hi[0] = a, lo[0] = b;
  split_f16(v.y, a, b), hi[1] = a, lo[1] = b;
  split_f16(v.z, a, b), hi[2] = a, lo[2] = b;
  split_f16(v.w, a, b), hi[3] = a, lo[3] = b;
  half_t* yp = y + row * 8 * K4 + x3_col(c);
  *(h4*)yp = hi;
  *(h4*)(yp + 32) = lo;
}

int split_rows_x3(const float* x, half_t* y, long M, int K, hipStream_t s) {
  if (M <= 0 || K <= 0 || (K % 32) || (((uintptr_t)x | (uintptr_t)y) & 15)) return SAMPT_ERR_ARG;
  const long n4 = M * (K / 4);
  hipLaunchKernelGGL(k_split_rows_x3, dim3(cdiv(n4, 256)), dim3(256), 0, s, (const float4*)x, y, n4, K / 4);
  SAMPT_CHECK_LAUNCH("split_rows_x3");
  return SAMPT_OK;
}

int cast_f32_f16(const float* x, half_t* y, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_cast_f32_f16, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, n);
  SAMPT_CHECK_LAUNCH("cast_f32_f16");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// SAM preprocess + 16x16 patch im2col (uint8 frames -> GEMM A operand)
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ void k_sam_patchify(const uint8_t* __restrict__ frames, int B, int H, int W, int img, int P, int chw,
                               float m0, float m1, float m2, float s0, float s1, float s2, TO* __restrict__ A) {
  // one thread per (patch row-of-P pixels): writes P contiguous k entries
  const int g = img / P;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * g * g * 3 * P;
  if (i >= total) return;
  int ky = (int)(i % P);
  long r = i / P;
  int c = (int)(r % 3);
  r /= 3;
  long patch = r;  // b*g*g + py*g + px
  int px = (int)(patch % g);
  int py = (int)((patch / g) % g);
  long b = patch / ((long)g * g);
  int y = py * P + ky;
  float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
  float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  TO* o = A + patch * (3 * P * P) + c * P * P + ky * P;
  for (int kx = 0; kx < P; ++kx) {
    int x = px * P + kx;
    float v = 0.f;
    if (y < H && x < W) {
      uint8_t u = chw ? frames[((b * 3 + c) * H + y) * (long)W + x] : frames[((b * H + y) * (long)W + x) * 3 + c];
      v = ((float)u - mean) / sd;
    }
    o[kx] = (TO)v;
  }
}

int sam_patchify(const uint8_t* frames, int chw, int B, int H, int W, int img, int P, const float* mean,
                 const float* stdv, void* A, int out_f16, hipStream_t s) {
  if (H > img || W > img || img % P) return SAMPT_ERR_ARG;
  int g = img / P;
  long total = (long)B * g * g * 3 * P;
  dim3 grid(cdiv(total, 256)), block(256);
  if (out_f16)
    hipLaunchKernelGGL(k_sam_patchify<half_t>, grid, block, 0, s, frames, B, H, W, img, P, chw, mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2], (half_t*)A);
  else
    hipLaunchKernelGGL(k_sam_patchify<float>, grid, block, 0, s, frames, B, H, W, img, P, chw, mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2], (float*)A);
  SAMPT_CHECK_LAUNCH("sam_patchify");
  return SAMPT_OK;
}

}  // namespace sampt

// =============================================================================================
// VOS post-processing: logits resize to target_hw (sam_pt.py:205-206) and the background-stack
// softmax/argmax over objects (vos_eval/eval.py:304, 326, 355; demo.py:140)
// =============================================================================================
namespace sampt {

// bilinear, align_corners=False, single channel, batch of n images (torch F.interpolate semantics)
__global__ void k_resize_logits(const float* __restrict__ src, int sh, int sw, float* __restrict__ dst, int dh, int dw,
                                long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % dw);
  long r = i / dw;
  int y = (int)(r % dh);
  long n = r / dh;
  int y0, y1, x0, x1;
  float ly, lx;
  bilinear_src(y, sh, dh, 0, y0, y1, ly);
  bilinear_src(x, sw, dw, 0, x0, x1, lx);
  const float* b = src + n * sh * sw;
  float v00 = b[(long)y0 * sw + x0], v01 = b[(long)y0 * sw + x1], v10 = b[(long)y1 * sw + x0], v11 = b[(long)y1 * sw + x1];
  float hy = 1.f - ly, hx = 1.f - lx;
  dst[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

int resize_logits(const float* src, int n, int sh, int sw, float* dst, int dh, int dw, hipStream_t s) {
  long total = (long)n * dh * dw;
  hipLaunchKernelGGL(k_resize_logits, dim3(cdiv(total, 256)), dim3(256), 0, s, src, sh, sw, dst, dh, dw, total);
  SAMPT_CHECK_LAUNCH("resize_logits");
  return SAMPT_OK;
}

// index mask: argmax over {background logit 0, object logits 1..M} == argmax(softmax(cat(0, logits))) (softmax is
// monotonic; first maximum wins like torch.argmax; -inf objects never win).  A NaN logit — bilinear resizing of a rejected
// (-inf) frame gives 0 * -inf on its border rows/columns, sam_pt.py:205-206 — makes the reference's whole softmax row NaN
// and torch.argmax of an all-NaN row is index 0: such pixels are background.
__global__ void k_index_masks(const float* __restrict__ logits, int M, long npix, uint8_t* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  float best = 0.f;
  int arg = 0;
  bool nan = false;
  for (int m = 0; m < M; ++m) {
    float v = logits[(long)m * npix + i];
    nan |= (v != v);
    if (v > best) {
      best = v;
      arg = m + 1;
    }
  }
  out[i] = (uint8_t)(nan ? 0 : arg);
}

int index_masks(const float* logits, int M, long npix, uint8_t* out, hipStream_t s) {
  if (M <= 0 || M > 254) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_index_masks, dim3(cdiv(npix, 256)), dim3(256), 0, s, logits, M, npix, out);
  SAMPT_CHECK_LAUNCH("index_masks");
  return SAMPT_OK;
}

// VOS variant (vos_eval/eval.py:318-326): object m's logits are -1e8 on the frames before its query frame qt[m], and on
// the query frame itself they are replaced by the ground-truth mask (+1e8 inside, -1e8 outside) before the argmax
__global__ void k_vos_index_masks(const float* __restrict__ logits, int M, int T, long hw, const int* __restrict__ qt,
                                  const uint8_t* __restrict__ gt, uint8_t* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)T * hw) return;
  const int t = (int)(i / hw);
  const long p = i - (long)t * hw;
  float best = 0.f;
  int arg = 0;
  bool nan = false;                 // after the overrides, as the evaluator applies them before the softmax
  for (int m = 0; m < M; ++m) {
    float v = logits[((long)m * T + t) * hw + p];
    if (t < qt[m]) v = -1e8f;
    else if (t == qt[m] && gt) v = gt[(long)m * hw + p] ? 1e8f : -1e8f;
    nan |= (v != v);
    if (v > best) {
      best = v;
      arg = m + 1;
    }
  }
  out[i] = (uint8_t)(nan ? 0 : arg);
}

// The evaluator's full tail for frames that were processed at another resolution (eval.py:326, 340-356): softmax over
// {background, objects} at the processing resolution (h, w), bilinear resize of the PROBABILITIES to (oh, ow)
// (F.interpolate, align_corners=False), argmax.  One thread per output pixel; M <= 32 objects.
__global__ void k_vos_index_masks_resized(const float* __restrict__ logits, int M, int T, int h, int w,
                                          const int* __restrict__ qt, const uint8_t* __restrict__ gt, int oh, int ow,
                                          uint8_t* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)T * oh * ow) return;
  const int x = (int)(i % ow), y = (int)((i / ow) % oh), t = (int)(i / ((long)ow * oh));
  int y0, y1, x0, x1;
  float ly, lx;
  {
    float sy = ((float)h / (float)oh) * ((float)y + 0.5f) - 0.5f, sx = ((float)w / (float)ow) * ((float)x + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy, sx = sx < 0.f ? 0.f : sx;
    y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
    y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    ly = sy - (float)y0, lx = sx - (float)x0;
  }
  const float wgt[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
  const int py[4] = {y0, y0, y1, y1}, px[4] = {x0, x1, x0, x1};
  float prob[33];
#pragma unroll 1
  for (int m = 0; m <= M; ++m) prob[m] = 0.f;
  const long hw = (long)h * w;
  for (int c = 0; c < 4; ++c) {
    const long p = (long)py[c] * w + px[c];
    float v[33];
    v[0] = 0.f;
    float mx = 0.f;
    for (int m = 0; m < M; ++m) {
      float a = logits[((long)m * T + t) * hw + p];
      if (t < qt[m]) a = -1e8f;
      else if (t == qt[m] && gt) a = gt[(long)m * hw + p] ? 1e8f : -1e8f;
      v[m + 1] = a;
      mx = fmaxf(mx, a);
    }
    float sum = 0.f;
    for (int m = 0; m <= M; ++m) {
      v[m] = expf(v[m] - mx);
      sum += v[m];
    }
    for (int m = 0; m <= M; ++m) prob[m] += wgt[c] * (v[m] / sum);
  }
  int arg = 0;
  float best = prob[0];
  for (int m = 1; m <= M; ++m)
    if (prob[m] > best) best = prob[m], arg = m;
  out[i] = (uint8_t)arg;
}

int vos_index_masks_resized(const float* logits, int M, int T, int h, int w, const int* qt, const uint8_t* gt, int oh,
                            int ow, uint8_t* out, hipStream_t s) {
  if (M <= 0 || M > 32 || !qt || oh <= 0 || ow <= 0) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_vos_index_masks_resized, dim3(cdiv((long)T * oh * ow, 256)), dim3(256), 0, s, logits, M, T, h, w, qt,
                     gt, oh, ow, out);
  SAMPT_CHECK_LAUNCH("vos_index_masks_resized");
  return SAMPT_OK;
}

int vos_index_masks(const float* logits, int M, int T, long hw, const int* qt, const uint8_t* gt, uint8_t* out,
                    hipStream_t s) {
  if (M <= 0 || M > 254 || !qt) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_vos_index_masks, dim3(cdiv((long)T * hw, 256)), dim3(256), 0, s, logits, M, T, hw, qt, gt, out);
  SAMPT_CHECK_LAUNCH("vos_index_masks");
  return SAMPT_OK;
}

// One pass of PIL's 8-bit separable resampler (ImagingResampleHorizontal/Vertical_8bpc): out = clip8((2^21 + sum_x
// in[xmin + x] * k[x]) >> 22) along one axis, intermediate image rounded to uint8 between the passes exactly as PIL does.
// The fixed-point coefficient tables (precompute_coeffs + normalize_coeffs_8bpc) come from the host.
// src viewed as [outer][in_len][inner] bytes, dst [outer][out_len][inner].
__global__ void k_pil_resample_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long outer, int in_len,
                                  int out_len, int inner, const int* __restrict__ coef, const int* __restrict__ bounds,
                                  int ksize) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= outer * out_len * inner) return;
  const int c = (int)(i % inner);
  const int xx = (int)((i / inner) % out_len);
  const long o = i / ((long)inner * out_len);
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = coef + (long)xx * ksize;
  const uint8_t* p = src + (o * in_len + xmin) * inner + c;
  int ss = 1 << 21;
  for (int x = 0; x < n; ++x) ss += (int)p[(long)x * inner] * k[x];
  ss >>= 22;
  dst[i] = (uint8_t)min(max(ss, 0), 255);
}

int pil_resample_u8(const uint8_t* src, uint8_t* dst, long outer, int in_len, int out_len, int inner, const int* coef,
                    const int* bounds, int ksize, hipStream_t s) {
  const long n = outer * out_len * inner;
  if (n <= 0 || ksize <= 0) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_pil_resample_u8, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, outer, in_len, out_len, inner, coef,
                     bounds, ksize);
  SAMPT_CHECK_LAUNCH("pil_resample_u8");
  return SAMPT_OK;
}

// rows[i] of a [*][N] matrix := bias (the qkv of SAM's zero-padded window tokens is the bias alone, App. A-3)
template <typename T>
__global__ void k_fill_rows_bias(T* __restrict__ out, const int* __restrict__ rows, const float* __restrict__ bias, int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < N) out[(long)rows[blockIdx.y] * N + c] = (T)bias[c];
}

// the same for a matrix of x3 rows (common.h): [*][2N] halves
__global__ void k_fill_rows_bias_x3(half_t* __restrict__ out, const int* __restrict__ rows, const float* __restrict__ bias,
                                    int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  half_t hi, lo;
  split_f16(bias[c], hi, lo);
  half_t* op = out + (long)rows[blockIdx.y] * 2 * N + x3_col(c);
  op[0] = hi, op[32] = lo;
}

// f16: 0 = f32 rows, 1 = fp16 rows, 2 = x3 rows
int fill_rows_bias(void* out, int f16, const int* rows, int nrows, const float* bias, int N, hipStream_t s) {
  if (nrows <= 0) return SAMPT_OK;
  dim3 grid(cdiv(N, 256), nrows);
  if (f16 == 2) hipLaunchKernelGGL(k_fill_rows_bias_x3, grid, dim3(256), 0, s, (half_t*)out, rows, bias, N);
  else if (f16) hipLaunchKernelGGL(k_fill_rows_bias<half_t>, grid, dim3(256), 0, s, (half_t*)out, rows, bias, N);
  else hipLaunchKernelGGL(k_fill_rows_bias<float>, grid, dim3(256), 0, s, (float*)out, rows, bias, N);
  SAMPT_CHECK_LAUNCH("fill_rows_bias");
  return SAMPT_OK;
}

}  // namespace sampt
