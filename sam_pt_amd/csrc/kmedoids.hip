// K-medoids query-point selection on the device.
//
// Reference: sam_pt/utils/query_points.py:62-99 — extract_kmedoid_points draws <= 1800 mask pixels and calls
// sklearn_extra.cluster.KMedoids(n_clusters).fit (third-party, absent; defaults: euclidean metric, method "alternate",
// init "heuristic", max_iter 300), restated on the host in sam_pt_amd/query_points.py:kmedoids_alternate.  These kernels
// reproduce that restatement BIT FOR BIT, which pins three things:
//   * distances are fp64: sqrt of an exact integer (pixel coordinates are integers), correctly rounded;
//   * every sum is numpy's pairwise summation of a contiguous fp64 row (np.sum(axis=1): 0.0 + pairwise_sum, blocks of
//     <= 128 elements with 8 strided accumulators, halves cut at multiples of 8 — numpy/core/src/umath/loops_utils.h.src),
//     over the same elements in the same order (members of a cluster in increasing index order, as np.where yields them);
//   * ties break towards the first index (np.argmin), a medoid moves only on a strictly smaller cost.
// The heuristic initialisation is np.argpartition of the row sums, whose output ORDER is an artefact of numpy's introselect:
// k_kmedoids_rowsums produces the sums, the host partitions n <= 1800 doubles, k_kmedoids_alternate runs the iterations.
//
// k_kmedoids_alternate is ONE workgroup of 1024 threads per point set: coordinates, labels, the ordered member list and the
// per-member costs live in LDS; per iteration every cluster costs (members)^2 distance evaluations spread over the threads
// (n = 1800, 8 clusters: ~400 per thread).  Latency, not throughput: the host restatement takes 60 - 160 ms per mask.
#include "ops.h"

namespace sampt {

namespace {
constexpr int KM_MAXN = 2048, KM_THREADS = 1024, KM_MAXK = 64;

__device__ __forceinline__ double km_dist(float ay, float ax, float by, float bx) {
  const double dy = (double)ay - (double)by, dx = (double)ax - (double)bx;
  return sqrt(fmax(dy * dy + dx * dx, 0.0));
}

// numpy's pairwise_sum over elem(0) .. elem(n - 1)
template <typename F>
__device__ double km_pairwise(const F& elem, int lo, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += elem(lo + i);
    return res;
  }
  if (n <= 128) {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = elem(lo + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += elem(lo + i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += elem(lo + i);
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  const double a = km_pairwise(elem, lo, n2);
  const double b = km_pairwise(elem, lo + n2, n - n2);
  return a + b;
}
}  // namespace

// out[i] = sum_j D[i][j]   (np.sum(D, axis=1))
__global__ __launch_bounds__(256) void k_kmedoids_rowsums(const float* __restrict__ xy, int n, double* __restrict__ out) {
  __shared__ float2 pts[KM_MAXN];
  for (int i = threadIdx.x; i < n; i += blockDim.x) pts[i] = ((const float2*)xy)[i];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 me = pts[i];
  out[i] = 0.0 + km_pairwise([&](int j) { return km_dist(me.x, me.y, pts[j].x, pts[j].y); }, 0, n);
}

__global__ __launch_bounds__(KM_THREADS) void k_kmedoids_alternate(const float* __restrict__ xy, int n, int K,
                                                                    int* __restrict__ medoids, int max_iter,
                                                                    int* __restrict__ iters_out) {
  __shared__ float2 pts[KM_MAXN];
  __shared__ short label[KM_MAXN];
  __shared__ int member[KM_MAXN];
  __shared__ double cost[KM_MAXN];
  __shared__ int med[KM_MAXK], old[KM_MAXK];
  __shared__ int wave_tot[KM_THREADS / 64];
  __shared__ double red_v[KM_THREADS / 64];
  __shared__ int red_i[KM_THREADS / 64];
  __shared__ int s_count, s_cur, s_changed;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n; i += KM_THREADS) pts[i] = ((const float2*)xy)[i];
  if (tid < K) med[tid] = medoids[tid];
  __syncthreads();
  const int per = (n + KM_THREADS - 1) / KM_THREADS;     // contiguous indices per thread (ordered compaction)
  int it = 0;
  for (; it < max_iter; ++it) {
    if (tid < K) old[tid] = med[tid];
    // labels = argmin over the medoids (first minimum)
    for (int j = tid; j < n; j += KM_THREADS) {
      const float2 p = pts[j];
      double best = km_dist(pts[med[0]].x, pts[med[0]].y, p.x, p.y);
      int bk = 0;
      for (int k = 1; k < K; ++k) {
        const double d = km_dist(pts[med[k]].x, pts[med[k]].y, p.x, p.y);
        if (d < best) best = d, bk = k;
      }
      label[j] = (short)bk;
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
      // ---- ordered member list of cluster k
      const int j0 = tid * per, j1 = min(j0 + per, n);
      int cnt = 0;
      for (int j = j0; j < j1; ++j) cnt += label[j] == k;
      int inc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      if (lane == 63) wave_tot[wave] = inc;
      __syncthreads();
      int base = 0;
      for (int w = 0; w < wave; ++w) base += wave_tot[w];
      if (tid == KM_THREADS - 1) s_count = base + inc, s_cur = 0;   // (np.argmax of an all-False match is 0)
      int pos = base + inc - cnt;
      for (int j = j0; j < j1; ++j)
        if (label[j] == k) member[pos++] = j;
      __syncthreads();
      const int m = s_count;
      if (m > 0) {
        // ---- cost of every member as the cluster's medoid: pairwise row sum over the members
        for (int i = tid; i < m; i += KM_THREADS) {
          const float2 me = pts[member[i]];
          cost[i] = 0.0 + km_pairwise([&](int j) { const float2 q = pts[member[j]]; return km_dist(me.x, me.y, q.x, q.y); }, 0, m);
          if (member[i] == med[k]) s_cur = i;
        }
        __syncthreads();
        // ---- first minimum
        double bv = INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < m; i += KM_THREADS)
          if (cost[i] < bv || (cost[i] == bv && i < bi)) bv = cost[i], bi = i;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const double ov = __shfl_xor(bv, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov < bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
        }
        if (lane == 0) red_v[wave] = bv, red_i[wave] = bi;
        __syncthreads();
        if (tid == 0) {
          for (int w = 1; w < KM_THREADS / 64; ++w)
            if (red_v[w] < bv || (red_v[w] == bv && red_i[w] < bi)) bv = red_v[w], bi = red_i[w];
          if (bv < cost[s_cur]) med[k] = member[bi];
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      int ch = 0;
      for (int k = 0; k < K; ++k) ch |= old[k] != med[k];
      s_changed = ch;
    }
    __syncthreads();
    if (!s_changed) {
      ++it;
      break;
    }
  }
  if (tid < K) medoids[tid] = med[tid];
  if (tid == 0 && iters_out) *iters_out = it;
}

int kmedoids_rowsums(const float* xy, int n, double* out, hipStream_t s) {
  if (!xy || !out || n <= 0 || n > KM_MAXN) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_kmedoids_rowsums, dim3(cdiv(n, 256)), dim3(256), 0, s, xy, n, out);
  SAMPT_CHECK_LAUNCH("kmedoids_rowsums");
  return SAMPT_OK;
}

int kmedoids_alternate(const float* xy, int n, int K, int* medoids, int max_iter, int* iters_out, hipStream_t s) {
  if (!xy || !medoids || n <= 0 || n > KM_MAXN || K <= 0 || K > KM_MAXK || K > n) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_kmedoids_alternate, dim3(1), dim3(KM_THREADS), 0, s, xy, n, K, medoids, max_iter, iters_out);
  SAMPT_CHECK_LAUNCH("kmedoids_alternate");
  return SAMPT_OK;
}

}  // namespace sampt
