// Shi-Tomasi query points and mask erosion on the device.
//
// Reference: sam_pt/utils/query_points.py:102-162 (extract_corner_points: cv2.cvtColor RGB2GRAY -> the mask eroded by a share
// of its bounding-box diagonal (:165-194, cv2.erode with a k x k kernel of ones; 6 % -> 2 % -> 1 % -> no erosion until >= 10
// pixels survive) -> cv2.goodFeaturesToTrack(maxCorners = n, qualityLevel = 0.001, minDistance = diagonal / n, mask = eroded)).
// OpenCV is a third-party dependency that is absent; sam_pt_amd/query_points.py restates its published algorithms in numpy
// (_rgb_to_gray_u8, _erode, corner_min_eigen_val, good_features_to_track) and these kernels reproduce THAT restatement bit
// for bit, which pins:
//   * gray = (9798 R + 19235 G + 3735 B + 2^14) >> 15 (integer);
//   * erosion = AND over the k x k window anchored at (k / 2, k / 2), pixels outside the image count as set (separable: rows,
//     then columns); k = int(double(float32 diagonal) * share), k = 0 -> 3, k = 1 -> identity — the arithmetic of
//     erode_mask_proportional_to_its_furthest_points_distance, evaluated by every thread from the mask's bounding box;
//   * min-eigenvalue map in float32 with numpy's operation order and NO fused multiply-adds (the file is compiled with fp
//     contraction off): Sobel 3 x 3 on the BORDER_REFLECT_101 image times 1 / (4 * 3 * 255), the three products, an
//     unnormalised 3 x 3 box sum over the reflect-101 PRODUCT maps accumulated as Python's sum() does (0 + t00 + t01 + ... + t22),
//     (a + c) - sqrt((a - c)^2 + b^2) with a correctly rounded square root (through fp64: 53 >= 2 * 24 + 2 bits);
//   * the threshold eig > float32(max over the mask) * float32(0.001), 3 x 3 local maxima (ties count), the one-pixel image
//     border excluded, candidates inside the eroded mask only;
//   * the greedy selection "strongest first, ties towards the HIGHER address, skip anything closer than minDistance to an
//     accepted corner" as n rounds of a masked arg-max over 64-bit keys (ordered float bits << 32 | address): a candidate once
//     too close to an accepted corner stays rejected, so the next accepted corner is the best remaining candidate that is far
//     from all accepted ones — the same sequence OpenCV's sorted scan with its cell grid produces (the grid only prunes the
//     distance tests: two corners closer than minDistance are always in adjacent cells of size round(minDistance)).
//     Distances are exact integers compared with (diagonal / n)^2 in fp64, as the host does.
// One launch sequence, no host round trip inside: the caller downloads n corners + a count at the end.
#pragma clang fp contract(off)
#include "ops.h"

namespace sampt {

namespace {

// state shared by the launch sequence (device memory, 16 int32):
//   [0..3] bbox of the mask: ymin, ymax, xmin, xmax      [4] pixels in the mask
//   [5..8] bbox of the CURRENT eroded mask               [9] pixels in it         [10] k of the current erosion (-1: mask itself)
//   [11] 1 = the current eroded mask is final            [12] number of corners found
//   [13] ordered-float bits of the max eigenvalue over the eroded mask
enum { QS_BB = 0, QS_CNT = 4, QS_EBB = 5, QS_ECNT = 9, QS_K = 10, QS_FINAL = 11, QS_NFOUND = 12, QS_MAX = 13, QS_NCAND = 14, QS_INTS = 16 };

__device__ __forceinline__ unsigned ord_bits(float v) {          // monotone map float -> uint32
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_float(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// torch.norm(px.max(0) - px.min(0)).item(): float32 sqrt of an exact integer, handed on as a double
__device__ __forceinline__ double bbox_diameter(const int* bb) {
  const float dy = (float)(bb[1] - bb[0]), dx = (float)(bb[3] - bb[2]);
  return (double)(float)sqrt((double)(dy * dy + dx * dx));
}

__global__ void k_qp_state_init(int* st) {
  const int i = threadIdx.x;
  if (i >= QS_INTS) return;
  int v = 0;
  if (i == QS_BB || i == QS_BB + 2 || i == QS_EBB || i == QS_EBB + 2) v = 0x7fffffff;
  if (i == QS_BB + 1 || i == QS_BB + 3 || i == QS_EBB + 1 || i == QS_EBB + 3) v = -1;
  if (i == QS_K) v = -1;
  st[i] = v;
}

// bounding box + pixel count of a {0, 1} byte mask (integer atomics: order-independent results)
__global__ __launch_bounds__(256) void k_qp_bbox(const uint8_t* __restrict__ m, int H, int W, int* st, int base, int gate) {
  if (gate && st[QS_FINAL]) return;
  const long n = (long)H * W;
  int ymin = 0x7fffffff, ymax = -1, xmin = 0x7fffffff, xmax = -1, cnt = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (m[i]) {
      const int y = (int)(i / W), x = (int)(i - (long)y * W);
      ymin = min(ymin, y), ymax = max(ymax, y), xmin = min(xmin, x), xmax = max(xmax, x), ++cnt;
    }
  for (int o = 32; o; o >>= 1) {
    ymin = min(ymin, __shfl_xor(ymin, o)), ymax = max(ymax, __shfl_xor(ymax, o));
    xmin = min(xmin, __shfl_xor(xmin, o)), xmax = max(xmax, __shfl_xor(xmax, o));
    cnt += __shfl_xor(cnt, o);
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicMin(st + base, ymin), atomicMax(st + base + 1, ymax), atomicMin(st + base + 2, xmin), atomicMax(st + base + 3, xmax);
    atomicAdd(st + base + 4, cnt);
  }
}

// start of an erosion attempt with `share` of the mask's diagonal: skipped once an earlier attempt left >= 10 pixels
__global__ void k_qp_erode_begin(int* st, double share) {
  if (st[QS_FINAL]) return;
  if (st[QS_K] >= 0 && st[QS_ECNT] >= 10) {        // the previous attempt stands
    st[QS_FINAL] = 1;
    return;
  }
  int k = (int)(bbox_diameter(st + QS_BB) * share);
  if (k == 0) k = 3;                               // cv2.erode with an empty kernel = the default 3 x 3
  st[QS_K] = k;
  st[QS_EBB] = 0x7fffffff, st[QS_EBB + 1] = -1, st[QS_EBB + 2] = 0x7fffffff, st[QS_EBB + 3] = -1, st[QS_ECNT] = 0;
}

// one axis of the separable erosion: out = AND of k consecutive pixels starting at (pos - k / 2), outside = set
__global__ __launch_bounds__(256) void k_qp_erode_axis(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W,
                                                       const int* st, int k_fixed, int horizontal) {
  int k = k_fixed;
  if (st) {
    if (st[QS_FINAL]) return;
    k = st[QS_K];
  }
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  const int a = k / 2;
  uint8_t v = 1;
  if (horizontal) {
    const int lo = max(x - a, 0), hi = min(x - a + k - 1, W - 1);
    for (int xx = lo; xx <= hi; ++xx) v &= in[(long)y * W + xx];
  } else {
    const int lo = max(y - a, 0), hi = min(y - a + k - 1, H - 1);
    for (int yy = lo; yy <= hi; ++yy) v &= in[(long)yy * W + x];
  }
  out[i] = v;
}

// after the three attempts: fewer than 10 pixels left -> the mask itself (query_points.py:127-128)
__global__ __launch_bounds__(256) void k_qp_erode_end(const uint8_t* __restrict__ mask, uint8_t* __restrict__ er, int H, int W,
                                                      int* st) {
  const bool keep = st[QS_ECNT] >= 10;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (!keep && i < (long)H * W) er[i] = mask[i] ? 1 : 0;
}
__global__ void k_qp_erode_end_state(int* st) {
  if (st[QS_ECNT] < 10) {
    for (int j = 0; j < 5; ++j) st[QS_EBB + j] = st[QS_BB + j];
    st[QS_K] = -1;
  }
  st[QS_FINAL] = 1;
}

__global__ __launch_bounds__(256) void k_qp_gray(const uint8_t* __restrict__ img, long hw, uint8_t* __restrict__ g) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  const int r = img[i], gg = img[hw + i], b = img[2 * hw + i];
  g[i] = (uint8_t)((r * 9798 + gg * 19235 + b * 3735 + (1 << 14)) >> 15);
}

__device__ __forceinline__ int refl(int p, int n) {               // BORDER_REFLECT_101 for p in [-2, n + 1], n >= 2
  if (p < 0) p = -p;
  if (p >= n) p = 2 * n - 2 - p;
  return p;
}

// cv::cornerMinEigenVal(blockSize 3, ksize 3) as query_points.corner_min_eigen_val evaluates it
__global__ __launch_bounds__(256) void k_qp_min_eig(const uint8_t* __restrict__ g, int H, int W, float* __restrict__ eig) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  const float scale = (float)(1.0 / (4.0 * 3.0 * 255.0));
  float sxx = 0.f, sxy = 0.f, syy = 0.f;                           // Python's sum(): 0 + t00 + t01 + ...
#pragma unroll
  for (int di = 0; di < 3; ++di) {
    const int yy = refl(y + di - 1, H);
    const int r0 = refl(yy - 1, H), r2 = refl(yy + 1, H);
#pragma unroll
    for (int dj = 0; dj < 3; ++dj) {
      const int xx = refl(x + dj - 1, W);
      const int c0 = refl(xx - 1, W), c2 = refl(xx + 1, W);
      const float g00 = (float)g[(long)r0 * W + c0], g01 = (float)g[(long)r0 * W + xx], g02 = (float)g[(long)r0 * W + c2];
      const float g10 = (float)g[(long)yy * W + c0], g12 = (float)g[(long)yy * W + c2];
      const float g20 = (float)g[(long)r2 * W + c0], g21 = (float)g[(long)r2 * W + xx], g22 = (float)g[(long)r2 * W + c2];
      // every intermediate below is a small integer: exact in float32 whatever the order
      const float dx = ((g02 - g00) + 2.f * (g12 - g10) + (g22 - g20)) * scale;
      const float dy = ((g20 - g00) + 2.f * (g21 - g01) + (g22 - g02)) * scale;
      const float pxx = dx * dx, pxy = dx * dy, pyy = dy * dy;
      sxx = sxx + pxx, sxy = sxy + pxy, syy = syy + pyy;
    }
  }
  const float a = sxx * 0.5f, b = sxy, c = syy * 0.5f;
  const float d = a - c;
  const float t = d * d + b * b;                                   // contraction is off: two roundings, then the sum
  const float s = (float)sqrt((double)t);                          // correctly rounded float32 square root
  eig[i] = (a + c) - s;
}

__global__ __launch_bounds__(256) void k_qp_max_masked(const float* __restrict__ eig, const uint8_t* __restrict__ m, long n, int* st) {
  unsigned best = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (m[i]) best = max(best, ord_bits(eig[i]));
  for (int o = 32; o; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
  if ((threadIdx.x & 63) == 0 && best) atomicMax((unsigned*)(st + QS_MAX), best);
}

// candidate keys: THRESH_TOZERO at max * quality, 3 x 3 local maximum, inside the eroded mask, off the image border
// ... and, compacted: the non-zero keys appended to `list` (capacity `cap`; st[QS_NCAND] counts every candidate, so a count above the
// capacity tells the greedy kernel to scan the dense array instead).  The order of the list is arbitrary — each greedy round takes the
// maximum key, and a key carries its position, so the result does not depend on it.
__global__ __launch_bounds__(256) void k_qp_candidates(const float* __restrict__ eig, const uint8_t* __restrict__ m, int H, int W,
                                                       int* st, float quality, unsigned long long* __restrict__ keys,
                                                       unsigned long long* __restrict__ list, int cap) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  unsigned long long key = 0;
  const unsigned mb = (unsigned)st[QS_MAX];
  if (mb && m[i] && y > 0 && y < H - 1 && x > 0 && x < W - 1) {
    const float thr = ord_float(mb) * quality;
    auto tz = [&](int yy, int xx) {
      const float v = eig[(long)yy * W + xx];
      return v > thr ? v : 0.f;
    };
    const float v = tz(y, x);
    if (v != 0.f) {
      float dil = v;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) dil = fmaxf(dil, tz(y + dy, x + dx));
      if (v == dil) key = ((unsigned long long)ord_bits(v) << 32) | (unsigned)i;
    }
  }
  keys[i] = key;
  if (key) {
    const int slot = atomicAdd(st + QS_NCAND, 1);
    if (slot < cap) list[slot] = key;
  }
}

// n rounds of the masked arg-max, ONE workgroup of 1024 threads, over the compacted candidate list (a frame has a few hundred to a few
// thousand candidates; the dense array — H * W keys, nearly all zero, re-read on every round — only when the list overflowed)
__global__ __launch_bounds__(1024) void k_qp_greedy(const unsigned long long* __restrict__ keys_dense, const unsigned long long* __restrict__ list,
                                                    int cap, int H, int W, int* st, int n_points, float* __restrict__ out_xy) {
  __shared__ unsigned long long red[16];
  __shared__ int ax[64], ay[64];
  __shared__ int n_acc;
  const bool dense = st[QS_NCAND] > cap;
  const unsigned long long* __restrict__ keys = dense ? keys_dense : list;
  const long n = dense ? (long)H * W : (long)st[QS_NCAND];
  const double md = bbox_diameter(st + QS_EBB) / (double)n_points;
  const bool check = !(md < 1.0);
  const double md2 = md * md;
  if (threadIdx.x == 0) n_acc = 0;
  __syncthreads();
  for (int r = 0; r < n_points; ++r) {
    const int na = n_acc;
    unsigned long long best = 0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long k = keys[i];
      if (k <= best) continue;
      const unsigned pos = (unsigned)(k & 0xffffffffu);                    // (the key's low word is the pixel index)
      const int y = (int)(pos / (unsigned)W), x = (int)(pos - (unsigned)y * (unsigned)W);
      bool good = true;
      for (int j = 0; j < na; ++j) {
        const long dx = x - ax[j], dy = y - ay[j];
        if (dx == 0 && dy == 0) { good = false; break; }          // already accepted
        if (check && (double)(dx * dx + dy * dy) < md2) { good = false; break; }
      }
      if (good) best = k;
    }
    for (int o = 32; o; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o);
      best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long b = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) b = red[w] > b ? red[w] : b;
      if (b) {
        const unsigned addr = (unsigned)(b & 0xffffffffu);
        const int y = (int)(addr / (unsigned)W), x = (int)(addr - (unsigned)y * (unsigned)W);
        ax[na] = x, ay[na] = y;
        out_xy[2 * na] = (float)x, out_xy[2 * na + 1] = (float)y;
        n_acc = na + 1;
      }
      red[0] = b;
    }
    __syncthreads();
    const bool found = red[0] != 0;
    __syncthreads();
    if (!found) break;
  }
  if (threadIdx.x == 0) st[QS_NFOUND] = n_acc;
}

}  // namespace

size_t qp_corners_workspace_bytes(int H, int W) {
  const size_t hw = (size_t)H * W;
  // state | eroded | tmp | gray (bytes, each padded to 256) | eig f32 | keys u64
  const size_t pad = (hw + 255) / 256 * 256;
  return 256 + 3 * pad + hw * 4 + 256 + hw * 8;
}

int qp_erode(const uint8_t* mask, int H, int W, int k, uint8_t* tmp, uint8_t* out, hipStream_t s) {
  if (!mask || !tmp || !out || H <= 0 || W <= 0 || k < 0) return SAMPT_ERR_ARG;
  if (k == 0) k = 3;
  const long n = (long)H * W;
  const dim3 grid(cdiv(n, 256)), block(256);
  hipLaunchKernelGGL(k_qp_erode_axis, grid, block, 0, s, mask, tmp, H, W, (const int*)nullptr, k, 1);
  hipLaunchKernelGGL(k_qp_erode_axis, grid, block, 0, s, (const uint8_t*)tmp, out, H, W, (const int*)nullptr, k, 0);
  SAMPT_CHECK_LAUNCH("qp_erode");
  return SAMPT_OK;
}

int qp_corners(const uint8_t* image, const uint8_t* mask, int H, int W, int n_points, float quality, float* out_xy, int* out_info,
               void* ws, size_t ws_bytes, hipStream_t s) {
  if (!image || !mask || !out_xy || !out_info || !ws || H < 3 || W < 3 || n_points < 1 || n_points > 64) return SAMPT_ERR_ARG;
  if (ws_bytes < qp_corners_workspace_bytes(H, W)) return SAMPT_ERR_ARG;
  const long n = (long)H * W;
  const size_t pad = ((size_t)n + 255) / 256 * 256;
  char* p = (char*)ws;
  int* st = (int*)p;
  uint8_t* er = (uint8_t*)(p + 256);
  uint8_t* tmp = er + pad;
  uint8_t* gray = tmp + pad;
  float* eig = (float*)(gray + pad);
  unsigned long long* keys = (unsigned long long*)((((uintptr_t)((char*)eig + (size_t)n * 4)) + 255) & ~(uintptr_t)255);
  const dim3 grid(cdiv(n, 256)), block(256), rgrid(256);
  hipLaunchKernelGGL(k_qp_state_init, dim3(1), dim3(64), 0, s, st);
  hipLaunchKernelGGL(k_qp_bbox, rgrid, block, 0, s, mask, H, W, st, (int)QS_BB, 0);
  const double shares[3] = {0.06, 0.02, 0.01};                       // query_points.py:123-126
  for (int a = 0; a < 3; ++a) {
    hipLaunchKernelGGL(k_qp_erode_begin, dim3(1), dim3(1), 0, s, st, shares[a]);
    hipLaunchKernelGGL(k_qp_erode_axis, grid, block, 0, s, mask, tmp, H, W, (const int*)st, 0, 1);
    hipLaunchKernelGGL(k_qp_erode_axis, grid, block, 0, s, (const uint8_t*)tmp, er, H, W, (const int*)st, 0, 0);
    hipLaunchKernelGGL(k_qp_bbox, rgrid, block, 0, s, (const uint8_t*)er, H, W, st, (int)QS_EBB, 1);
  }
  hipLaunchKernelGGL(k_qp_erode_end, grid, block, 0, s, mask, er, H, W, st);
  hipLaunchKernelGGL(k_qp_erode_end_state, dim3(1), dim3(1), 0, s, st);
  hipLaunchKernelGGL(k_qp_gray, grid, block, 0, s, image, n, gray);
  hipLaunchKernelGGL(k_qp_min_eig, grid, block, 0, s, (const uint8_t*)gray, H, W, eig);
  hipLaunchKernelGGL(k_qp_max_masked, rgrid, block, 0, s, (const float*)eig, (const uint8_t*)er, n, st);
  // (the candidate list lives in tmp | gray, both dead by now: 2 * pad bytes)
  unsigned long long* list = (unsigned long long*)tmp;
  const int cap = (int)(2 * pad / 8);
  hipLaunchKernelGGL(k_qp_candidates, grid, block, 0, s, (const float*)eig, (const uint8_t*)er, H, W, st, quality, keys, list, cap);
  hipLaunchKernelGGL(k_qp_greedy, dim3(1), dim3(1024), 0, s, (const unsigned long long*)keys, (const unsigned long long*)list, cap, H, W, st,
                     n_points, out_xy);
  SAMPT_CHECK_LAUNCH("qp_corners");
  if (hipMemcpyAsync(out_info, st, QS_INTS * sizeof(int), hipMemcpyDeviceToDevice, s) != hipSuccess) return SAMPT_ERR_HIP;
  return SAMPT_OK;
}

}  // namespace sampt
