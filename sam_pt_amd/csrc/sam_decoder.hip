// Prompt encoder / mask decoder helper kernels (all fp32).
// Every kernel carries a frame-batch dimension F: the per-(frame, object) prompt chains of one clip are independent,
// so the engine runs pass r of ALL frames as one launch (tokens [F][Nt][256], image tokens [F][4096][256]) instead of
// F latency-bound single-frame launches.
#include "ops.h"

namespace sampt {

// ---------------------------------------------------------------------------------------------
// decoder token matrix [F][Nt][256]: rows 0..n_out-1 = iou token + 4 mask tokens (+ HQ token), then the sparse tokens:
// k points (random-Fourier PE + label embedding), then (no box) one "not a point" pad row | (box) two corner rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_sam_tokens(const float* __restrict__ out_tokens, const float* __restrict__ pts,
                                                    const int* __restrict__ labels, int k, int ld_pts,
                                                    const float* __restrict__ box, const float* __restrict__ gauss,
                                                    const float* __restrict__ point_emb,
                                                    const float* __restrict__ not_a_point, float img_size, int Nt,
                                                    int n_out, const int* __restrict__ k_item,
                                                    int* __restrict__ ntok, float* __restrict__ tokens) {
  const int trow = blockIdx.x, f = blockIdx.y, j = threadIdx.x;
  float* out = tokens + ((long)f * Nt + trow) * 256;
  // ragged batch: item f uses its first k_item[f] points; its tokens are packed at the front, the rest of its rows are
  // zero padding that the attention kernels mask out as keys (ntok[f] = valid token count)
  if (k_item) k = min(k, k_item[f]);
  const int nvalid = n_out + k + (box ? 2 : 1);
  if (ntok && trow == 0 && j == 0) ntok[f] = nvalid;
  if (trow >= nvalid) {
    out[j] = 0.f, out[128 + j] = 0.f;
    return;
  }
  if (trow < n_out) {
    out[j] = out_tokens[trow * 256 + j];
    out[128 + j] = out_tokens[trow * 256 + 128 + j];
    return;
  }
  const int row = trow - n_out;
  float x, y;
  int label;
  const float* emb;
  if (row < k) {
    x = pts[((long)f * ld_pts + row) * 2], y = pts[((long)f * ld_pts + row) * 2 + 1];
    label = labels[(long)f * ld_pts + row];
    emb = label == 0 ? point_emb : (label == 1 ? point_emb + 256 : not_a_point);
  } else if (box == nullptr) {
    x = 0.f, y = 0.f, label = -1, emb = not_a_point;
  } else {
    int c = row - k;  // corner 0 = (x0,y0), corner 1 = (x1,y1)
    x = box[f * 4 + c * 2], y = box[f * 4 + c * 2 + 1];
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
  out[j] = s + emb[j];
  out[128 + j] = c + emb[128 + j];
}

int sam_tokens(const float* out_tokens, int n_out, const float* pts, const int* labels, int k, int ld_pts,
               const float* box, const float* gauss, const float* point_emb, const float* not_a_point, float img_size,
               int F, const int* k_item, int* ntok, float* tokens, hipStream_t s) {
  int Nt = n_out + k + (box ? 2 : 1);
  hipLaunchKernelGGL(k_sam_tokens, dim3(Nt, F), dim3(128), 0, s, out_tokens, pts, labels, k, ld_pts, box, gauss,
                     point_emb, not_a_point, img_size, Nt, n_out, k_item, ntok, tokens);
  SAMPT_CHECK_LAUNCH("sam_tokens");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// attention with many keys and few queries: one workgroup per (QPW queries, head, frame); scores stay in registers.
// K and V of a (frame, head) are streamed once per workgroup and shared by its QPW queries: with one query per
// workgroup the token->image attention re-read them Nq times through L2 (6 TB/s of L2 traffic, L2-bandwidth bound).
// Per-query arithmetic and reduction order do not depend on QPW.
// ---------------------------------------------------------------------------------------------
template <int HD, int QPW>
__global__ __launch_bounds__(256) void k_attn_rowblock(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ out, int Nq,
                                                       int Nk, int ld, const int* __restrict__ nk_item) {
  constexpr int KPT = 16;  // keys per thread (Nk <= 4096)
  __shared__ float red[QPW][8];
  __shared__ float accs[QPW][4][HD];
  const int q0 = blockIdx.x * QPW, h = blockIdx.y, f = blockIdx.z;
  q += (long)f * Nq * ld, out += (long)f * Nq * ld;
  k += (long)f * Nk * ld, v += (long)f * Nk * ld;
  if (nk_item) Nk = nk_item[f];   // ragged batch: only the item's valid tokens are keys (strides keep the padded Nk)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float qv[QPW][HD];
#pragma unroll
  for (int j = 0; j < QPW; ++j) {
    const int qi = min(q0 + j, Nq - 1);      // surplus queries of the last workgroup recompute the last one (not stored)
#pragma unroll
    for (int c = 0; c < HD; ++c) qv[j][c] = q[(long)qi * ld + h * HD + c];
  }
  const float inv = sqrtf((float)HD);
  float sc[QPW][KPT];
  float m[QPW];
#pragma unroll
  for (int j = 0; j < QPW; ++j) m[j] = -INFINITY;
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    int key = tid + 256 * i;
#pragma unroll
    for (int j = 0; j < QPW; ++j) sc[j][i] = -INFINITY;
    if (key < Nk) {
      const float4* kp = (const float4*)(k + (long)key * ld + h * HD);
      float a[QPW];
#pragma unroll
      for (int j = 0; j < QPW; ++j) a[j] = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        float4 kk = kp[c];
#pragma unroll
        for (int j = 0; j < QPW; ++j)
          a[j] += qv[j][4 * c] * kk.x + qv[j][4 * c + 1] * kk.y + qv[j][4 * c + 2] * kk.z + qv[j][4 * c + 3] * kk.w;
      }
#pragma unroll
      for (int j = 0; j < QPW; ++j) {
        sc[j][i] = a[j] / inv;
        m[j] = fmaxf(m[j], sc[j][i]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < QPW; ++j) {
    m[j] = wave_max(m[j]);
    if (lane == 0) red[j][wave] = m[j];
  }
  __syncthreads();
  float acc[QPW][HD], sum[QPW];
#pragma unroll
  for (int j = 0; j < QPW; ++j) {
    m[j] = fmaxf(fmaxf(red[j][0], red[j][1]), fmaxf(red[j][2], red[j][3]));
    sum[j] = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[j][c] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    int key = tid + 256 * i;
    if (key < Nk) {
      float p[QPW];
#pragma unroll
      for (int j = 0; j < QPW; ++j) {
        p[j] = expf(sc[j][i] - m[j]);
        sum[j] += p[j];
      }
      const float4* vp = (const float4*)(v + (long)key * ld + h * HD);
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        float4 vv = vp[c];
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
          acc[j][4 * c] += p[j] * vv.x;
          acc[j][4 * c + 1] += p[j] * vv.y;
          acc[j][4 * c + 2] += p[j] * vv.z;
          acc[j][4 * c + 3] += p[j] * vv.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < QPW; ++j) {
    float t = wave_sum(sum[j]);
    if (lane == 0) red[j][4 + wave] = t;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      float a = wave_sum(acc[j][c]);
      if (lane == 0) accs[j][wave][c] = a;
    }
  }
  __syncthreads();
  if (tid < HD * QPW) {
    const int j = tid / HD, c = tid - j * HD;
    if (q0 + j < Nq) {
      float tot = (red[j][4] + red[j][5]) + (red[j][6] + red[j][7]);
      float a = (accs[j][0][c] + accs[j][1][c]) + (accs[j][2][c] + accs[j][3][c]);
      out[(long)(q0 + j) * ld + h * HD + c] = a / tot;
    }
  }
}

// Self-attention among the prompt tokens (8 heads x 32 channels, 7 .. a few hundred tokens): one wave per (64 queries,
// head, item), lane = query.  K and V rows of a (head, item) are wave-uniform, so they are read with scalar loads and used
// as SGPR operands (k_attn_rowblock spent a 256-thread workgroup and two block reductions per query: 236 us per launch at
// 87 tokens x 32 items).
template <int HD>
__global__ __launch_bounds__(256) void k_attn_tokens_s(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ out, int Nq,
                                                       int Nk, int ld, const int* __restrict__ nk_item, float rscale) {
  // 4 waves share the 64 queries of the workgroup and split the keys (key % 4 == wave): the loop is a chain of scalar-load
  // latencies, so four short chains beat one long one; maxima, sums and accumulators meet in LDS
  __shared__ float s_max[4][64];
  __shared__ float s_acc[4][64][HD + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y, f = blockIdx.z;
  const int qi = blockIdx.x * 64 + lane;
  const long qrow = ((long)f * Nq + min(qi, Nq - 1)) * ld + h * HD;
  float qv[HD];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const float4 t = *(const float4*)(q + qrow + c * 4);
    qv[4 * c] = t.x, qv[4 * c + 1] = t.y, qv[4 * c + 2] = t.z, qv[4 * c + 3] = t.w;
  }
  typedef const __attribute__((address_space(4))) float* cptr;     // constant address space: scalar loads (see above)
  const cptr kh = (cptr)(uintptr_t)(k + (long)f * Nk * ld + h * HD);
  const cptr vh = (cptr)(uintptr_t)(v + (long)f * Nk * ld + h * HD);
  const int nvalid = __builtin_amdgcn_readfirstlane(nk_item ? nk_item[f] : Nk);
  float m = -INFINITY;
  for (int key = w; key < nvalid; key += 4) {
    const cptr kp = kh + (long)key * ld;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) a += qv[c] * kp[c];
    m = fmaxf(m, a * rscale);
  }
  s_max[w][lane] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_max[0][lane], s_max[1][lane]), fmaxf(s_max[2][lane], s_max[3][lane]));
  float sum = 0.f, acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = 0.f;
  for (int key = w; key < nvalid; key += 4) {
    const cptr kp = kh + (long)key * ld;
    const cptr vp = vh + (long)key * ld;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) a += qv[c] * kp[c];
    const float p = expf(a * rscale - m);
    sum += p;
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] += p * vp[c];
  }
#pragma unroll
  for (int c = 0; c < HD; ++c) s_acc[w][lane][c] = acc[c];
  s_acc[w][lane][HD] = sum;
  __syncthreads();
  if (qi >= Nq) return;
  const float tot = (s_acc[0][lane][HD] + s_acc[1][lane][HD]) + (s_acc[2][lane][HD] + s_acc[3][lane][HD]);
  constexpr int CW = HD / 4;                                        // channels finished by each wave
  float o[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int cc = w * CW + c;
    o[c] = ((s_acc[0][lane][cc] + s_acc[1][lane][cc]) + (s_acc[2][lane][cc] + s_acc[3][lane][cc])) / tot;
  }
#pragma unroll
  for (int c = 0; c < CW / 4; ++c)
    *(float4*)(out + qrow + w * CW + c * 4) = make_float4(o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
}

int attn_rowblock(const float* q, const float* k, const float* v, float* out, int F, int Nq, int Nk, int heads, int hd,
                  const int* nk_item, hipStream_t s) {
  if (Nk > 4096 || Nk <= 0 || Nq <= 0 || F <= 0) return SAMPT_ERR_ARG;
  int ld = heads * hd;
  if (hd == 32 && ld % 16 == 0 && !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15)) {
    hipLaunchKernelGGL((k_attn_tokens_s<32>), dim3(cdiv(Nq, 64), heads, F), dim3(256), 0, s, q, k, v, out, Nq, Nk, ld, nk_item,
                       1.0f / sqrtf(32.0f));
    SAMPT_CHECK_LAUNCH("attn_tokens_s");
    return SAMPT_OK;
  }
  if (hd == 16 && Nk > 256) {            // token -> image: long key streams, share them between 4 queries
    hipLaunchKernelGGL((k_attn_rowblock<16, 4>), dim3(cdiv(Nq, 4), heads, F), dim3(256), 0, s, q, k, v, out, Nq, Nk, ld,
                       nk_item);
  } else if (hd == 16) {
    hipLaunchKernelGGL((k_attn_rowblock<16, 1>), dim3(Nq, heads, F), dim3(256), 0, s, q, k, v, out, Nq, Nk, ld, nk_item);
  } else if (hd == 32) {
    hipLaunchKernelGGL((k_attn_rowblock<32, 1>), dim3(Nq, heads, F), dim3(256), 0, s, q, k, v, out, Nq, Nk, ld, nk_item);
  } else {
    return SAMPT_ERR_UNSUPPORTED;
  }
  SAMPT_CHECK_LAUNCH("attn_rowblock");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// token -> image attention of the two-way transformer (8 heads x 16 channels, 10-20 prompt tokens against the 4096 image
// tokens of every frame): HBM-bound on K and V, which every frame reads exactly once.
//   * the keys of a frame are split over KS workgroups (F alone would leave most CUs idle), each handling up to 16
//     queries and ALL heads: a wave's lane owns one float4 of a 512-byte K/V row (lane -> half-wave = key parity, head =
//     (lane & 31) >> 2, quarter = lane & 3), so every global load is a fully coalesced 1 KiB and nothing is re-read per
//     head or per query (k_attn_rowblock re-read each row through L2 once per head and per 4 queries, 64 bytes at a time);
//   * flash-style running softmax per (query, head) over chunks of 4 keys; the quarter-dots are summed with two DPP
//     row shuffles; partial (max, sum, acc) states go to a workspace and k_attn_t2i_merge combines the KS splits.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int T2I_NQ = 8;           // queries per workgroup (their running softmax states live in registers).  (SAM-PT's 13 - 21 prompt
                                    // tokens are two blocks, so K / V are streamed twice; 16 per workgroup — one stream — was measured in
                                    // round 6: 256 VGPRs, two waves per SIMD, 78 us against 46: the kernel is bound by its exp / FMA work
                                    // per (query, key) and the occupancy that hides its loads, not by the bytes; profiles/r6_c39_*)
constexpr int T2I_REC = 128 + 16;   // floats per (split, query): acc[128] + max[8] + sum[8]
constexpr float T2I_SCALE = 0.25f * 1.4426950408889634f;   // scores are kept in log2 units: softmax via v_exp_f32 alone
#define T2I_EXP(x) __builtin_amdgcn_exp2f(x)
constexpr float T2I_MIN = -1e30f;   // "no key yet" running maximum (finite: exp(T2I_MIN - m) = 0 without inf - inf)

__device__ __forceinline__ float quad_sum(float v) {      // sum over the 4 lanes of a quad, result in all 4 (DPP, no LDS)
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  return v;
}
}  // namespace

__global__ __launch_bounds__(256) void k_attn_t2i_part(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ part,
                                                       float* __restrict__ out, int Nq, int Nk, int kper, int ldkv) {
  constexpr int NQ = T2I_NQ;
  __shared__ float4 s_acc[8][NQ][32];   // one partial state per (wave, key parity)
  __shared__ float4 s_q[NQ][32];        // the queries stay in LDS (broadcast reads) instead of 4 NQ registers per lane
  __shared__ float s_m[8][NQ][8], s_s[8][NQ][8];
  const int ks = blockIdx.x, qb = blockIdx.y, f = blockIdx.z, KS = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, hp = lane & 31;
  q += (long)f * Nq * 128, k += (long)f * Nk * ldkv, v += (long)f * Nk * ldkv;   // k / v rows: 128 floats, stride ldkv
  const int q0 = qb * NQ;
  for (int e = tid; e < NQ * 32; e += 256)
    s_q[e >> 5][e & 31] = *(const float4*)(q + (long)min(q0 + (e >> 5), Nq - 1) * 128 + (e & 31) * 4);
  __syncthreads();
  float m[NQ], sum[NQ];
  float4 acc[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) m[j] = T2I_MIN, sum[j] = 0.f, acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int k0 = ks * kper, k1 = min(Nk, k0 + kper);
  // a wave takes 8 consecutive keys per step (4 per half-wave): rows k0 + 32 i + 8 wave + {0..7}; byte offsets inside
  // a frame's K / V fit 32 bits (checked by the launcher)
  const char* kc = (const char*)k;
  const char* vc = (const char*)v;
  auto load = [&](int kb, float4* kk, float4* vv) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int key = kb + 2 * c + half;
      const unsigned off = (unsigned)(key < k1 ? key : k0) * ((unsigned)ldkv * 4u) + (unsigned)hp * 16u;
      kk[c] = *(const float4*)(kc + off);
      vv[c] = *(const float4*)(vc + off);
    }
  };
  auto step = [&](int kb, const float4* kk, const float4* vv) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      float sc[4];
      const float4 qj = s_q[j][hp];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float d = qj.x * kk[c].x + qj.y * kk[c].y + qj.z * kk[c].z + qj.w * kk[c].w;
        d = quad_sum(d) * T2I_SCALE;                          // / sqrt(16), in the base-2 domain of T2I_EXP
        sc[c] = kb + 2 * c + half < k1 ? d : -INFINITY;
      }
      const float mn = fmaxf(fmaxf(m[j], fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));
      const float r = T2I_EXP(m[j] - mn);
      m[j] = mn;
      float s_ = sum[j] * r;
      float4 a = make_float4(acc[j].x * r, acc[j].y * r, acc[j].z * r, acc[j].w * r);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float pr = T2I_EXP(sc[c] - mn);
        s_ += pr;
        a.x += pr * vv[c].x, a.y += pr * vv[c].y, a.z += pr * vv[c].z, a.w += pr * vv[c].w;
      }
      sum[j] = s_, acc[j] = a;
    }
  };
  // (two resident waves per SIMD overlap one wave's loads with the other's arithmetic; an explicit register ping-pong
  //  only made hipcc re-order the loads against the in-order vmcnt counter)
  for (int kb = k0 + wave * 8; kb < k1; kb += 32) {
    float4 kk[4], vv[4];
    load(kb, kk, vv);
    step(kb, kk, vv);
  }
  // ---- merge the 8 partial streams of the workgroup through LDS
  const int st_ = wave * 2 + half;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    s_acc[st_][j][hp] = acc[j];
    if ((hp & 3) == 0) s_m[st_][j][hp >> 2] = m[j], s_s[st_][j][hp >> 2] = sum[j];
  }
  __syncthreads();
  for (int e = tid; e < NQ * 32; e += 256) {
    const int j = e >> 5, c4 = e & 31, h = c4 >> 2;
    if (q0 + j >= Nq) continue;
    float mn = T2I_MIN;
#pragma unroll
    for (int w = 0; w < 8; ++w) mn = fmaxf(mn, s_m[w][j][h]);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float st = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float r = T2I_EXP(s_m[w][j][h] - mn);
      const float4 t = s_acc[w][j][c4];
      a.x += t.x * r, a.y += t.y * r, a.z += t.z * r, a.w += t.w * r;
      st += s_s[w][j][h] * r;
    }
    if (KS == 1) {
      *(float4*)(out + ((long)f * Nq + q0 + j) * 128 + c4 * 4) = make_float4(a.x / st, a.y / st, a.z / st, a.w / st);
    } else {
      float* rec = part + (((long)f * Nq + q0 + j) * KS + ks) * T2I_REC;
      *(float4*)(rec + c4 * 4) = a;
      if ((c4 & 3) == 0) rec[128 + h] = mn, rec[136 + h] = st;
    }
  }
}

// out[f][q][h*16 + c] = sum_ks acc * exp(m_ks - M) / sum_ks sum * exp(m_ks - M): one thread per (f, q, float4)
__global__ void k_attn_t2i_merge(const float* __restrict__ part, float* __restrict__ out, long nrows, int KS) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nrows * 32) return;
  const long row = e >> 5;
  const int c4 = (int)(e & 31), h = c4 >> 2;
  const float* rec = part + row * KS * T2I_REC;
  float mn = T2I_MIN;
  for (int s = 0; s < KS; ++s) mn = fmaxf(mn, rec[(long)s * T2I_REC + 128 + h]);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  float st = 0.f;
  for (int s = 0; s < KS; ++s) {
    const float* rs = rec + (long)s * T2I_REC;
    const float r = T2I_EXP(rs[128 + h] - mn);
    const float4 t = *(const float4*)(rs + c4 * 4);
    a.x += t.x * r, a.y += t.y * r, a.z += t.z * r, a.w += t.w * r;
    st += rs[136 + h] * r;
  }
  *(float4*)(out + row * 128 + c4 * 4) = make_float4(a.x / st, a.y / st, a.z / st, a.w / st);
}

// key split: enough workgroups to fill the chip (>= ~512), at least 64 keys each, at most 32 splits
static void attn_t2i_plan(int F, int Nq, int Nk, int& qblocks, int& KS, int& kper) {
  qblocks = cdiv(Nq, T2I_NQ);
  const long wg = (long)F * qblocks;
  int want = (int)((512 + wg - 1) / wg);
  want = want < 1 ? 1 : (want > 32 ? 32 : want);
  const int most = Nk / 64 > 0 ? Nk / 64 : 1;
  if (want > most) want = most;
  kper = cdiv(cdiv(Nk, want), 32) * 32;
  KS = cdiv(Nk, kper);
}

size_t attn_t2i_workspace_floats(int F, int Nq, int Nk) {
  // F * Nq * KS records with KS <= 512 / (F * qblocks) + 1 and Nq / qblocks <= T2I_NQ: a bound that grows with F and Nq,
  // so a workspace sized for the largest batch / prompt also covers every smaller one
  (void)Nk;
  return ((size_t)512 * T2I_NQ + (size_t)F * Nq) * T2I_REC;
}

int attn_t2i(const float* q, const float* k, const float* v, float* out, int F, int Nq, int Nk, float* ws, size_t ws_floats,
             hipStream_t s, int ldkv) {
  if (Nk <= 0 || Nq <= 0 || F <= 0 || ldkv < 128 || ldkv % 4 || (long)Nk * ldkv * 4 >= (1L << 32)) return SAMPT_ERR_ARG;
  int qb, KS, kper;
  attn_t2i_plan(F, Nq, Nk, qb, KS, kper);
  if (KS > 1 && (!ws || ws_floats < (size_t)F * Nq * KS * T2I_REC)) return SAMPT_ERR_WORKSPACE;
  hipLaunchKernelGGL(k_attn_t2i_part, dim3(KS, qb, F), dim3(256), 0, s, q, k, v, ws, out, Nq, Nk, kper, ldkv);
  SAMPT_CHECK_LAUNCH("attn_t2i_part");
  if (KS > 1) {
    const long nrows = (long)F * Nq;
    hipLaunchKernelGGL(k_attn_t2i_merge, dim3((unsigned)cdiv(nrows * 32, 256)), dim3(256), 0, s, ws, out, nrows, KS);
    SAMPT_CHECK_LAUNCH("attn_t2i_merge");
  }
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// attention with few keys and many queries (image -> token): one thread per (query, head); K/V staged through LDS in
// chunks of <= `chunk` keys with a running (max, sum) softmax, so the number of prompt tokens is unbounded.  With
// Nk <= chunk (every SAM-PT prompt up to 120 points) there is exactly one chunk and no rescale.
// ---------------------------------------------------------------------------------------------
// STAGE (heads * HD == 128): the 32 query rows of a workgroup go through LDS both ways — global loads and stores are whole
// 512-byte rows, the (query, head) threads pick their 64-byte slices from LDS (direct slices put the 64 lanes of a load on
// 64 different cache lines, 16 bytes each: the kernel was bound by that, not by its arithmetic).
template <int HD, bool STAGE>
__global__ __launch_bounds__(256) void k_attn_fewkeys(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, float* __restrict__ out, int Nq,
                                                      int Nk, int heads, const int* __restrict__ nk_item, int chunk,
                                                      int ldq) {
  extern __shared__ float kv[];  // [2][chunk][ld]
  const int ld = heads * HD, f = blockIdx.y;
  q += (long)f * Nq * ldq, out += (long)f * Nq * ld;
  k += (long)f * Nk * ld, v += (long)f * Nk * ld;
  float* ks = kv;
  float* vs = kv + (long)chunk * ld;
  constexpr int SLD = 132;                        // staged row stride in floats (128 + 4: rows start on different banks)
  float* stage = kv + 2L * chunk * ld;            // STAGE: [32][SLD]
  const int nvalid = nk_item ? nk_item[f] : Nk;   // ragged batch: only the item's valid tokens are keys
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < (long)Nq * heads;
  const int qi = live ? (int)(idx / heads) : 0, h = live ? (int)(idx % heads) : 0;
  const int q0 = blockIdx.x * 32, ql = threadIdx.x >> 3;          // STAGE: first query of the workgroup, local query
  float qv[HD];
  if (STAGE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + i * 256, r = e >> 5, c4 = e & 31;
      *(float4*)(stage + r * SLD + c4 * 4) = *(const float4*)(q + (long)min(q0 + r, Nq - 1) * ldq + c4 * 4);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      float4 t = *(const float4*)(stage + ql * SLD + h * HD + c * 4);
      qv[4 * c] = t.x, qv[4 * c + 1] = t.y, qv[4 * c + 2] = t.z, qv[4 * c + 3] = t.w;
    }
  } else {
    const float4* qp = (const float4*)(q + (long)qi * ldq + h * HD);
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      float4 t = qp[c];
      qv[4 * c] = t.x, qv[4 * c + 1] = t.y, qv[4 * c + 2] = t.z, qv[4 * c + 3] = t.w;
    }
  }
  static_assert(HD == 16, "the score scale below is written as an exact reciprocal of sqrt(16)");
  const float rinv = 0.25f;        // 1 / sqrt(HD): a power of two, so the product equals the division bit for bit
  float m = -INFINITY, sum = 0.f;
  float acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = 0.f;
  for (int c0 = 0; c0 < nvalid; c0 += chunk) {
    const int n = min(chunk, nvalid - c0);
    if (c0) __syncthreads();
    {   // stage the chunk: 16-byte loads, up to 4 K and 4 V vectors in flight per thread before the first LDS store
      const float4* k4 = (const float4*)(k + (long)c0 * ld);
      const float4* v4 = (const float4*)(v + (long)c0 * ld);
      const int n4 = n * (ld >> 2);
      for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * 256) {
        float4 kr[4], vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * 256;
          if (i < n4) kr[u] = k4[i], vr[u] = v4[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * 256;
          if (i < n4) ((float4*)ks)[i] = kr[u], ((float4*)vs)[i] = vr[u];
        }
      }
    }
    __syncthreads();
    float mc = -INFINITY;
    for (int key = 0; key < n; ++key) {
      const float* kp = ks + key * ld + h * HD;
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) a += qv[c] * kp[c];
      mc = fmaxf(mc, a * rinv);
    }
    if (mc > m) {            // new running maximum: rescale what has been accumulated (never taken in the first chunk's
      if (m > -INFINITY) {   // accumulation order, so a single-chunk result is bit-identical to the unchunked kernel)
        const float r = expf(m - mc);
        sum *= r;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] *= r;
      }
      m = mc;
    }
    for (int key = 0; key < n; ++key) {
      const float* kp = ks + key * ld + h * HD;
      const float* vp = vs + key * ld + h * HD;
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) a += qv[c] * kp[c];
      float p = expf(a * rinv - m);
      sum += p;
#pragma unroll
      for (int c = 0; c < HD; ++c) acc[c] += p * vp[c];
    }
  }
  if (STAGE) {
    __syncthreads();                                // every thread has taken its query slice out of `stage`
#pragma unroll
    for (int c = 0; c < HD / 4; ++c)
      *(float4*)(stage + ql * SLD + h * HD + c * 4) =
          make_float4(acc[4 * c] / sum, acc[4 * c + 1] / sum, acc[4 * c + 2] / sum, acc[4 * c + 3] / sum);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + i * 256, r = e >> 5, c4 = e & 31;
      if (q0 + r < Nq) *(float4*)(out + (long)(q0 + r) * ld + c4 * 4) = *(const float4*)(stage + r * SLD + c4 * 4);
    }
    return;
  }
  if (!live) return;
  float4* op = (float4*)(out + (long)qi * ld + h * HD);
#pragma unroll
  for (int c = 0; c < HD / 4; ++c)
    op[c] = make_float4(acc[4 * c] / sum, acc[4 * c + 1] / sum, acc[4 * c + 2] / sum, acc[4 * c + 3] / sum);
}

// 8 heads x 16 channels (the image -> token attention of every SAM decoder): wave = head, lane = query.  The K / V slice
// of a (frame, head) is then the same for all 64 lanes of a wave, so it is read with SCALAR loads (s_load_dwordx16 per key
// row) and enters the FMAs as an SGPR operand: no LDS staging of the keys at all — the (query, head)-per-thread kernel
// above read every key from LDS once per thread and was bound by LDS bandwidth (1.1 ms per launch at 87 prompt tokens x
// 32 items).  Any number of keys, no chunking; the 64 query rows of a workgroup go through LDS both ways so that global
// loads and stores are whole 512-byte rows.
__global__ __launch_bounds__(512) void k_attn_fewkeys_s(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ out, int Nq,
                                                        int Nk, const int* __restrict__ nk_item, int ldq) {
  constexpr int HD = 16, LD = 128, SLD = 132;
  __shared__ float stage[64 * SLD];
  const int f = blockIdx.y, q0 = blockIdx.x * 64, tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6), ql = tid & 63;
  q += (long)f * Nq * ldq, out += (long)f * Nq * LD;
  // wave-uniform pointers in the constant address space: hipcc then always selects scalar loads for them (K and V are
  // written by earlier launches only)
  typedef const __attribute__((address_space(4))) float* cptr;
  const cptr kh = (cptr)(uintptr_t)(k + (long)f * Nk * LD + h * HD);
  const cptr vh = (cptr)(uintptr_t)(v + (long)f * Nk * LD + h * HD);
  // ragged batch: only the item's valid tokens are keys (uniform over the workgroup)
  const int nvalid = __builtin_amdgcn_readfirstlane(nk_item ? nk_item[f] : Nk);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * 512, r = e >> 5, c4 = e & 31;
    *(float4*)(stage + r * SLD + c4 * 4) = *(const float4*)(q + (long)min(q0 + r, Nq - 1) * ldq + c4 * 4);
  }
  __syncthreads();
  float qv[HD];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const float4 t = *(const float4*)(stage + ql * SLD + h * HD + c * 4);
    qv[4 * c] = t.x, qv[4 * c + 1] = t.y, qv[4 * c + 2] = t.z, qv[4 * c + 3] = t.w;
  }
  float m = -INFINITY;
  for (int key = 0; key < nvalid; ++key) {
    const cptr kp = kh + (long)key * LD;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) a += qv[c] * kp[c];
    m = fmaxf(m, a * 0.25f);                                         // / sqrt(16)
  }
  float sum = 0.f, acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = 0.f;
  for (int key = 0; key < nvalid; ++key) {
    const cptr kp = kh + (long)key * LD;
    const cptr vp = vh + (long)key * LD;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) a += qv[c] * kp[c];
    const float p = expf(a * 0.25f - m);
    sum += p;
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] += p * vp[c];
  }
  __syncthreads();                                  // every thread has taken its query slice out of `stage`
#pragma unroll
  for (int c = 0; c < HD / 4; ++c)
    *(float4*)(stage + ql * SLD + h * HD + c * 4) =
        make_float4(acc[4 * c] / sum, acc[4 * c + 1] / sum, acc[4 * c + 2] / sum, acc[4 * c + 3] / sum);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * 512, r = e >> 5, c4 = e & 31;
    if (q0 + r < Nq) *(float4*)(out + (long)(q0 + r) * LD + c4 * 4) = *(const float4*)(stage + r * SLD + c4 * 4);
  }
}

int attn_fewkeys(const float* q, const float* k, const float* v, float* out, int F, int Nq, int Nk, int heads, int hd,
                 const int* nk_item, hipStream_t s, int ldq) {
  if (Nk <= 0 || hd != 16 || F <= 0) return SAMPT_ERR_UNSUPPORTED;
  if (ldq == 0) ldq = heads * hd;
  if (ldq < heads * hd || ldq % 4) return SAMPT_ERR_ARG;
  if (heads == 8 && hd == 16) {
    hipLaunchKernelGGL(k_attn_fewkeys_s, dim3(cdiv(Nq, 64), F), dim3(512), 0, s, q, k, v, out, Nq, Nk, nk_item, ldq);
    SAMPT_CHECK_LAUNCH("attn_fewkeys_s");
    return SAMPT_OK;
  }
  const int chunk = Nk < 128 ? Nk : 128;
  const bool stage = heads * hd == 128;
  size_t sh = ((size_t)2 * chunk * heads * hd + (stage ? 32 * 132 : 0)) * sizeof(float);
  if (sh > 64 * 1024) {  // above the default dynamic-LDS limit (gfx950 has 160 KiB per workgroup)
    static bool raised = false;
    if (!raised) {
      if (hipFuncSetAttribute((const void*)k_attn_fewkeys<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              152 * 1024) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_attn_fewkeys<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              144 * 1024) != hipSuccess)
        return SAMPT_ERR_HIP;
      raised = true;
    }
  }
  if (stage)
    hipLaunchKernelGGL((k_attn_fewkeys<16, true>), dim3(cdiv((long)Nq * heads, 256), F), dim3(256), sh, s, q, k, v, out, Nq,
                       Nk, heads, nk_item, chunk, ldq);
  else
    hipLaunchKernelGGL((k_attn_fewkeys<16, false>), dim3(cdiv((long)Nq * heads, 256), F), dim3(256), sh, s, q, k, v, out, Nq,
                       Nk, heads, nk_item, chunk, ldq);
  SAMPT_CHECK_LAUNCH("attn_fewkeys");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// low_res[f][p] = <hyper[f], upscaled[f][p]>   (masks = hyper_in @ upscaled, App. A-4)
// ---------------------------------------------------------------------------------------------
// HQ-SAM adds a second term  <hyper2[f], up2[f][p]>  (mask_sam + mask_hq), summed after each dot is complete.
__global__ void k_sam_mask_dot(const float* __restrict__ up, const float* __restrict__ hyper, int ld_hyper,
                               const float* __restrict__ up2, const float* __restrict__ hyper2, int ld_hyper2,
                               float* __restrict__ low, int npix, int C) {
  int p = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
  if (p >= npix) return;
  const float4* u = (const float4*)(up + ((long)f * npix + p) * C);
  const float* hy = hyper + (long)f * ld_hyper;
  float a = 0.f;
  for (int c = 0; c < C / 4; ++c) {
    float4 t = u[c];
    a += hy[4 * c] * t.x + hy[4 * c + 1] * t.y + hy[4 * c + 2] * t.z + hy[4 * c + 3] * t.w;
  }
  if (up2) {
    const float4* u2 = (const float4*)(up2 + ((long)f * npix + p) * C);
    const float* h2 = hyper2 + (long)f * ld_hyper2;
    float b = 0.f;
    for (int c = 0; c < C / 4; ++c) {
      float4 t = u2[c];
      b += h2[4 * c] * t.x + h2[4 * c + 1] * t.y + h2[4 * c + 2] * t.z + h2[4 * c + 3] * t.w;
    }
    a += b;
  }
  low[(long)f * npix + p] = a;
}

// C == 32 (every SAM variant: transformer_dim / 8): 8 lanes per pixel, one float4 each, so a wave reads 1 KiB of
// consecutive bytes per load (one thread per pixel strides the lanes 128 bytes apart: 4x slower); the 8 partial dots are
// summed with three DPP steps (quad_perm, quad_perm, row_half_mirror)
__device__ __forceinline__ float oct_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  return v;
}

__global__ void k_sam_mask_dot32(const float* __restrict__ up, const float* __restrict__ hyper, int ld_hyper,
                                 const float* __restrict__ up2, const float* __restrict__ hyper2, int ld_hyper2,
                                 float* __restrict__ low, int npix) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int f = blockIdx.y, part = (int)(idx & 7);
  const long p = idx >> 3;
  const bool live = p < npix;                      // (npix * 8 is a multiple of the block: whole octets are live or dead)
  const long pe = ((long)f * npix + (live ? p : 0)) * 32 + part * 4;
  const float4 t = *(const float4*)(up + pe);
  const float4 h = *(const float4*)(hyper + (long)f * ld_hyper + part * 4);
  // (explicit fused multiply-adds: the same chain as gemm_x3_wres.hip's fused epilogue, see k_layernorm_rows_d64)
  float a = oct_sum(__builtin_fmaf(h.w, t.w, __builtin_fmaf(h.z, t.z, __builtin_fmaf(h.y, t.y, h.x * t.x))));
  if (up2) {
    const float4 t2 = *(const float4*)(up2 + pe);
    const float4 h2 = *(const float4*)(hyper2 + (long)f * ld_hyper2 + part * 4);
    a += oct_sum(h2.x * t2.x + h2.y * t2.y + h2.z * t2.z + h2.w * t2.w);
  }
  if (live && part == 0) low[(long)f * npix + p] = a;
}

int sam_mask_dot(const float* up, const float* hyper, int ld_hyper, const float* up2, const float* hyper2, int ld_hyper2,
                 float* low_res, int F, int npix, int C, hipStream_t s) {
  if (C % 4) return SAMPT_ERR_ARG;
  if (C == 32 && ld_hyper % 4 == 0 && (!up2 || ld_hyper2 % 4 == 0) &&
      !(((uintptr_t)hyper | (uintptr_t)hyper2 | (uintptr_t)up | (uintptr_t)up2) & 15)) {
    hipLaunchKernelGGL(k_sam_mask_dot32, dim3(cdiv((long)npix * 8, 256), F), dim3(256), 0, s, up, hyper, ld_hyper, up2,
                       hyper2, ld_hyper2, low_res, npix);
    SAMPT_CHECK_LAUNCH("sam_mask_dot32");
    return SAMPT_OK;
  }
  hipLaunchKernelGGL(k_sam_mask_dot, dim3(cdiv(npix, 256), F), dim3(256), 0, s, up, hyper, ld_hyper, up2, hyper2,
                     ld_hyper2, low_res, npix, C);
  SAMPT_CHECK_LAUNCH("sam_mask_dot");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// Sam.postprocess_masks fused: low (LxL) --bilinear--> (img x img) --crop (in_h,in_w)--> bilinear --> (oh,ow),
// both align_corners=False.  Optionally reduces the bounding box / count of logits > 0 (sam_pt.py:809-820) in two
// deterministic stages (per-workgroup partials, then k_bbox_final): state int[5] = {xmin, ymin, xmax, ymax, count}.
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

__device__ __forceinline__ void bbox_block_reduce(bool pos, int x, int y, int (*red)[5], int* out5) {
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
    out5[0] = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
    out5[1] = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
    out5[2] = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
    out5[3] = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
    out5[4] = red[0][4] + red[1][4] + red[2][4] + red[3][4];
  }
}

__global__ __launch_bounds__(256) void k_sam_postprocess(const float* __restrict__ low, int L, int img, int in_h,
                                                         int in_w, float* __restrict__ out, int oh, int ow,
                                                         int* __restrict__ partial) {
  __shared__ int red[4][5];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
  low += (long)f * L * L;
  out += (long)f * oh * ow;
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
  if (partial) {
    const int nb = gridDim.x * gridDim.y;
    bbox_block_reduce(pos, x, y, red, partial + 5 * ((long)f * nb + blockIdx.y * gridDim.x + blockIdx.x));
  }
}

__global__ __launch_bounds__(256) void k_bbox_final(const int* __restrict__ partial, int nblocks, int* __restrict__ bbox,
                                                    int ld_bbox) {
  __shared__ int red[4][5];
  const int f = blockIdx.x;
  partial += (long)f * nblocks * 5;
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
    int* b = bbox + (long)f * ld_bbox;
    b[0] = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
    b[1] = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
    b[2] = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
    b[3] = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
    b[4] = red[0][4] + red[1][4] + red[2][4] + red[3][4];
  }
}

size_t bbox_partial_ints(int oh, int ow) { return (size_t)5 * cdiv(ow, 64) * cdiv(oh, 4); }

int sam_postprocess_bbox(const float* low, int L, int img, int in_h, int in_w, float* out, int oh, int ow, int F,
                         int* bbox, int* bbox_partial, hipStream_t s) {
  if (in_h > img || in_w > img || (bbox && !bbox_partial) || F <= 0) return SAMPT_ERR_ARG;
  dim3 grid(cdiv(ow, 64), cdiv(oh, 4), F);
  hipLaunchKernelGGL(k_sam_postprocess, grid, dim3(256), 0, s, low, L, img, in_h, in_w, out, oh, ow,
                     bbox ? bbox_partial : nullptr);
  SAMPT_CHECK_LAUNCH("sam_postprocess");
  if (bbox) {
    hipLaunchKernelGGL(k_bbox_final, dim3(F), dim3(256), 0, s, bbox_partial, (int)(grid.x * grid.y), bbox, 5);
    SAMPT_CHECK_LAUNCH("bbox_final");
  }
  return SAMPT_OK;
}

int sam_postprocess(const float* low, int L, int img, int in_h, int in_w, float* out, int oh, int ow, hipStream_t s) {
  return sam_postprocess_bbox(low, L, img, in_h, in_w, out, oh, ow, 1, nullptr, nullptr, s);
}

// standalone bbox of logits > 0 (one image)
__global__ __launch_bounds__(256) void k_bbox_from_logits(const float* __restrict__ logits, int h, int w,
                                                          int* __restrict__ partial) {
  __shared__ int red[4][5];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  bool pos = x < w && y < h && logits[(long)y * w + x] > 0.f;
  bbox_block_reduce(pos, x, y, red, partial + 5 * (blockIdx.y * gridDim.x + blockIdx.x));
}

int bbox_from_logits_state(const float* logits, int h, int w, int* bbox_state, int* bbox_partial, hipStream_t s) {
  dim3 grid(cdiv(w, 64), cdiv(h, 4));
  hipLaunchKernelGGL(k_bbox_from_logits, grid, dim3(256), 0, s, logits, h, w, bbox_partial);
  SAMPT_CHECK_LAUNCH("bbox_from_logits");
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, s, bbox_partial, (int)(grid.x * grid.y), bbox_state, 5);
  SAMPT_CHECK_LAUNCH("bbox_final");
  return SAMPT_OK;
}

// =============================================================================================
// mask-input embedding, refinement gating, commit and IoU-threshold finalisation
// =============================================================================================

// stage A: conv2x2 s2 (1 -> C1) + LayerNorm2d(C1) + GELU ; stage B: conv2x2 s2 (C1 -> C2) + LayerNorm2d + GELU
template <int CIN, int COUT>
__global__ void k_mask_down(const float* __restrict__ in, int ih, int iw, const float* __restrict__ w,
                            const float* __restrict__ b, const float* __restrict__ lnw, const float* __restrict__ lnb,
                            float* __restrict__ out) {
  // in: [F][ih][iw][CIN] NHWC ; out: [F][ih/2][iw/2][COUT] ; w: [COUT][CIN][2][2]
  int oh = ih / 2, ow = iw / 2;
  int p = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
  if (p >= oh * ow) return;
  in += (long)f * ih * iw * CIN;
  out += (long)f * oh * ow * COUT;
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

// stage C: src[f][p][c] = feat[f][p][c] + b2[c] + sum_k w2[c][k] * e[f][p][k]
constexpr int MEO_PIX = 32;
__global__ void k_mask_embed_out(const float* __restrict__ e, int C2, const float* __restrict__ w2,
                                 const float* __restrict__ b2, const float* __restrict__ feat, float* __restrict__ src,
                                 long npix_total) {
  // 256 threads = output channels; MEO_PIX pixels per workgroup (each thread loads its 16 weights once per workgroup: at
  // 4 pixels per workgroup the weight re-reads through L2 were 4x the kernel's HBM traffic)
  __shared__ float es[MEO_PIX][16];
  const int c = threadIdx.x;
  const long p0 = (long)blockIdx.x * MEO_PIX;
  for (int i = c; i < MEO_PIX * C2; i += 256) {
    long p = p0 + i / C2;
    es[i / C2][i % C2] = p < npix_total ? e[p * C2 + i % C2] : 0.f;
  }
  __syncthreads();
  float wr[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) wr[k] = k < C2 ? w2[c * C2 + k] : 0.f;
  const float bb = b2[c];
#pragma unroll 4
  for (int i = 0; i < MEO_PIX; ++i) {
    long p = p0 + i;
    if (p >= npix_total) break;
    float a = bb;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += wr[k] * es[i][k];
    src[p * 256 + c] = feat[p * 256 + c] + a;
  }
}

int sam_mask_embed_src(const float* mask, int g, int F, const MaskEmbedW& w, const float* feat, float* tmp0,
                       float* tmp1, float* src, hipStream_t s) {
  int L = 4 * g;
  hipLaunchKernelGGL((k_mask_down<1, 4>), dim3(cdiv((L / 2) * (L / 2), 256), F), dim3(256), 0, s, mask, L, L, w.w0,
                     w.b0, w.ln0w, w.ln0b, tmp0);
  SAMPT_CHECK_LAUNCH("mask_down0");
  hipLaunchKernelGGL((k_mask_down<4, 16>), dim3(cdiv(g * g, 256), F), dim3(256), 0, s, tmp0, L / 2, L / 2, w.w1, w.b1,
                     w.ln1w, w.ln1b, tmp1);
  SAMPT_CHECK_LAUNCH("mask_down1");
  long npix = (long)F * g * g;
  hipLaunchKernelGGL(k_mask_embed_out, dim3(cdiv(npix, MEO_PIX)), dim3(256), 0, s, tmp1, 16, w.w2, w.b2, feat, src, npix);
  SAMPT_CHECK_LAUNCH("mask_embed_out");
  return SAMPT_OK;
}

// active[f] &= count(bbox_cur[f]) >= 2 ; box_f[f] = bbox_cur[f]     (sam_pt.py:809-820).  first: every item starts active
// (no memset in front of the chain: the captured chain is kernel nodes only — a memset node of a replayed hipGraph was
// observed to land late on ROCm 7.2, re-activating items a later pass had switched off)
__global__ void k_sam_refine_gate(int* __restrict__ active, const int* __restrict__ bbox_cur, float* __restrict__ box_f,
                                  int F, int first) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int* b = bbox_cur + f * 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) box_f[f * 4 + i] = (float)b[i];
  active[f] = ((first || active[f] != 0) && b[4] >= 2) ? 1 : 0;
}

int sam_refine_gate(int* active, const int* bbox_cur, float* box_f, int F, bool first, hipStream_t s) {
  hipLaunchKernelGGL(k_sam_refine_gate, dim3(cdiv(F, 64)), dim3(64), 0, s, active, bbox_cur, box_f, F, first ? 1 : 0);
  SAMPT_CHECK_LAUNCH("sam_refine_gate");
  return SAMPT_OK;
}

__global__ void k_sam_commit(const int* __restrict__ active, const float* __restrict__ cand_logits,
                             float* __restrict__ cur_logits, long n_logits, const float* __restrict__ cand_low,
                             float* __restrict__ cur_low, long n_low, const float* __restrict__ cand_iou,
                             float* __restrict__ cur_iou, const int* __restrict__ cand_bbox, int* __restrict__ cur_bbox) {
  const int f = blockIdx.y;
  if (active[f] == 0) return;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_logits) cur_logits[f * n_logits + i] = cand_logits[f * n_logits + i];
  if (i < n_low) cur_low[f * n_low + i] = cand_low[f * n_low + i];
  if (i == 0) cur_iou[f] = cand_iou[f];
  if (i < 5) cur_bbox[f * 5 + i] = cand_bbox[f * 5 + i];
}

int sam_commit(const int* active, const float* cand_logits, float* cur_logits, long n_logits, const float* cand_low,
               float* cur_low, long n_low, const float* cand_iou, float* cur_iou, const int* cand_bbox, int* cur_bbox,
               int F, hipStream_t s) {
  long n = n_logits > n_low ? n_logits : n_low;
  hipLaunchKernelGGL(k_sam_commit, dim3(cdiv(n, 256), F), dim3(256), 0, s, active, cand_logits, cur_logits, n_logits,
                     cand_low, cur_low, n_low, cand_iou, cur_iou, cand_bbox, cur_bbox);
  SAMPT_CHECK_LAUNCH("sam_commit");
  return SAMPT_OK;
}

__global__ void k_sam_finalize_mask(const float* __restrict__ logits, const float* __restrict__ iou, float thr,
                                    float* __restrict__ out, float* __restrict__ score_out, long n) {
  const int f = blockIdx.y;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float sc = iou[f];
  if (i < n) out[f * n + i] = sc < thr ? -INFINITY : logits[f * n + i];
  if (i == 0) score_out[f] = sc;
}

int sam_finalize_mask(const float* logits, const float* iou, float thr, float* out, float* score_out, long n, int F,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_sam_finalize_mask, dim3(cdiv(n, 256), F), dim3(256), 0, s, logits, iou, thr, out, score_out, n);
  SAMPT_CHECK_LAUNCH("sam_finalize_mask");
  return SAMPT_OK;
}

}  // namespace sampt
