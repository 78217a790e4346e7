// SAM ViT attention on gfx950:
//   attn = softmax(q.k^T * scale + rel_h[q, kh] + rel_w[q, kw]) . v      (decomposed relative position bias)
//
//  * vit_rel_bias            — the two small bias tables per query (shared by both modes)
//  * softmax_rel_rows        — exact-fp32 mode: bias + softmax over a materialised score matrix
//  * vit_flash_attention_f16 — fused flash-style kernel: fp16 MFMA (32x32x16), fp32 online softmax; K / V tiles by LDS-DMA
//                              (double-buffered), V consumed through transposing LDS reads, bias tables built by MFMA in the
//                              prologue and kept in LDS, scores never leave registers.
#include "ops.h"

namespace sampt {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// rel tables: relhT[bh][kh][q] = <q_vec, rel_pos_h[qh - kh + S-1]>,  relwT[bh][kw][q] likewise with qw
// One workgroup per (grid row qh, head, batch): its S query vectors are staged in LDS as fp32.
// ---------------------------------------------------------------------------------------------
template <typename TQ>
__global__ __launch_bounds__(256) void k_vit_rel_bias(const TQ* __restrict__ qkv, const float* __restrict__ rel_h,
                                                      const float* __restrict__ rel_w, int S, int heads, int hd,
                                                      float* __restrict__ relhT, float* __restrict__ relwT) {
  extern __shared__ float qs[];  // [S][hd+1]
  const int qh = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int N = S * S, D = heads * hd, ldq = hd + 1;
  for (int i = threadIdx.x; i < S * hd; i += 256) {
    int t = i / hd, c = i - t * hd;
    qs[t * ldq + c] = (float)qkv[((long)(b * N + qh * S + t)) * 3 * D + h * hd + c];
  }
  __syncthreads();
  const long bh = (long)b * heads + h;
  for (int i = threadIdx.x; i < S * 2 * S; i += 256) {
    int kk = i / S, qw = i - kk * S;  // kk in [0, 2S): first S -> relh (key row), next S -> relw (key col)
    const float* tab;
    if (kk < S) tab = rel_h + (long)(qh - kk + S - 1) * hd;
    else tab = rel_w + (long)(qw - (kk - S) + S - 1) * hd;
    const float* qv = qs + qw * ldq;
    float a = 0.f;
    for (int c = 0; c < hd; ++c) a += qv[c] * tab[c];
    int q = qh * S + qw;
    if (kk < S) relhT[(bh * S + kk) * N + q] = a;
    else relwT[(bh * S + (kk - S)) * N + q] = a;
  }
}

int vit_rel_bias(const void* qkv, int qkv_f16, const float* rel_h, const float* rel_w, int B, int S, int heads,
                 int hd, float* relh, float* relw, hipStream_t s) {
  dim3 grid(S, heads, B), block(256);
  size_t sh = (size_t)S * (hd + 1) * sizeof(float);
  if (qkv_f16)
    hipLaunchKernelGGL(k_vit_rel_bias<half_t>, grid, block, sh, s, (const half_t*)qkv, rel_h, rel_w, S, heads, hd, relh,
                       relw);
  else
    hipLaunchKernelGGL(k_vit_rel_bias<float>, grid, block, sh, s, (const float*)qkv, rel_h, rel_w, S, heads, hd, relh,
                       relw);
  SAMPT_CHECK_LAUNCH("vit_rel_bias");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// exact mode: scores[bh][q][k] += relhT[bh][k/S][q] + relwT[bh][k%S][q]; softmax over k (in place)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_softmax_rel_rows(float* __restrict__ scores, const float* __restrict__ relhT,
                                                          const float* __restrict__ relwT, int N, int S) {
  __shared__ float red[8];
  const int q = blockIdx.x;
  const long bh = blockIdx.y;
  float* row = scores + (bh * N + q) * N;
  const float* rh = relhT + bh * S * N + q;
  const float* rw = relwT + bh * S * N + q;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = -INFINITY;
  for (int k = tid; k < N; k += 256) {
    float v = row[k] + rh[(long)(k / S) * N] + rw[(long)(k % S) * N];
    row[k] = v;
    m = fmaxf(m, v);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int k = tid; k < N; k += 256) {
    float e = expf(row[k] - m);
    row[k] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  for (int k = tid; k < N; k += 256) row[k] = row[k] / sum;
}

int softmax_rel_rows(float* scores, const float* relh, const float* relw, long BH, int Nq, int S, hipStream_t s) {
  if (Nq != S * S) return SAMPT_ERR_ARG;
  hipLaunchKernelGGL(k_softmax_rel_rows, dim3(Nq, (unsigned)BH), dim3(256), 0, s, scores, relh, relw, Nq, S);
  SAMPT_CHECK_LAUNCH("softmax_rel_rows");
  return SAMPT_OK;
}

// ---------------------------------------------------------------------------------------------
// fused flash attention, f16 MFMA 32x32x16.
//
// "Swapped" formulation: each wave owns 32 queries and computes S^T = K.Q^T (A = K tile rows, B = Q^T), so a
// lane holds 16 keys x ONE query column per 32x32 tile: softmax row statistics are per-lane (+1 exchange with
// lane^32) and the per-query rescale factors line up with O^T = V^T.P^T, whose columns are the same queries.
// P^T is fed to the second MFMA straight from the score registers (any k-slot permutation is legal as long as
// the A operand uses the same one, and V^T is gathered from LDS with exactly that permutation).
//
// K and V tiles go HBM -> LDS by LDS-DMA straight from the packed qkv rows into ONE buffer: no staging registers, no
// transposing stores; a workgroup exposes the DMA latency of every tile, the two other workgroups resident on its CU hide it.  V stays ROW-major
// ([key slot][channel], as it lies in HBM); the A operand of O^T = V^T.P^T is gathered with ds_read_b64_tr_b16, gfx950's
// transposing LDS read: within a 16-lane group, lane s passes the address of 4 consecutive channels of key (s >> 2), channel
// chunk (s & 3), and lane c receives channel c of the 4 keys (tools/probes/tr_probe.hip) — exactly the 4 consecutive key
// slots a lane's P fragment holds, twice per k-step.  An LDS image written by DMA is linear, so bank conflicts are handled
// on the source side: K rows (HD * 2 bytes) XOR their 16-byte chunk index with row bits (for HD = 80, 10 chunks: bit 0 with
// row bit 3); V rows are 192 B for HD = 80 (conflict-free for the 4-row x 64-byte footprint of a transposing read; the 32
// pad bytes come from a constant page whose first half is 1.0: channel HD of "V" is all ones, see LROW below) and for
// HD = 64 swap their 64-byte halves on rows with bit 1 set.
// ---------------------------------------------------------------------------------------------
__device__ half_t g_flash_pad[16] = {(half_t)1.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f,
                                     (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f,
                                     (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};

template <int HD, int NW, int SG>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 2) void k_flash_f16(const half_t* __restrict__ qkv, const float* __restrict__ rel_h,
                                                       const float* __restrict__ rel_w, half_t* __restrict__ out, int N,
                                                       int heads, float scale, FlashPad pad, int B, int uh) {
  constexpr int KS = HD / 16;            // k-steps of QK^T
  constexpr int DT = (HD + 31) / 32;     // 32-row tiles of O^T
  constexpr int QT = NW * 32;
  constexpr int NT = NW * 64;
  constexpr bool LROW = DT * 32 > HD;
  constexpr int VP = LROW ? DT * 32 : HD;   // halves per (row-major) V row in LDS
  constexpr int RLD = QT + 2;
  constexpr int CPR = HD / 8, CPV = VP / 8;   // 16-byte chunks per K row / per LDS V row
  // K chunk ^= (row >> KSH) & KSWZ
  constexpr int KSWZ = CPR == 8 ? 7 : (CPR == 4 ? 3 : (CPR == 10 ? 1 : 0)), KSH = CPR == 4 ? 2 : (CPR == 10 ? 3 : 1);
  // ONE K / V tile buffer, and the rel_w table — dead once its values sit in registers — aliased onto it: 39 KiB for 64 x 64
  // tokens / head dim 80, three workgroups per CU.  (Rounds 2 - 3 double-buffered the tiles and kept both tables: 77 KiB, two
  // workgroups per CU; occupancy beats the intra-workgroup overlap: global blocks 1407 -> 1227 us per 8 frames, windowed
  // unchanged, 112.4 -> 115.0 fps in an A / B of one call, profiles/r4_c7_*.  The opposite trade for the 14 x 14 windows — the
  // whole window's K / V resident, no DMA wait or barrier in the key loop, 78 KiB and two workgroups per CU — loses: 258 vs 228 us,
  // profiles/r4_c10_*; so do, for the windows at three workgroups per CU, the rel_w table in its own LDS with tile 0 requested
  // before the prologue (208 vs 210 us), a second tile buffer on top of that (213), and two-wave workgroups (231),
  // profiles/r4_c17_*.  What did move the windowed kernel: keeping a window's 32 workgroups on one XCD (233 -> 217 us, common.h
  // flash_wg_decode); what moved the global one: the DMA address table below (1087 -> 1041 us, profiles/r4_c18_*).)
  constexpr int KB = 64 * HD * 2, VB = 64 * VP * 2, RELB = SG * RLD * 2;
  constexpr int BUFB = KB + VB > RELB ? KB + VB : ((RELB + 1023) / 1024) * 1024;
  __shared__ __attribute__((aligned(1024))) char tile_s[BUFB];
  __shared__ half_t relh_s[SG][RLD];
  half_t(*Ks)[64][HD] = (half_t(*)[64][HD])tile_s;
  half_t(*Vs)[64][VP] = (half_t(*)[64][VP])(tile_s + KB);
  half_t(*relw_s)[RLD] = (half_t(*)[RLD])tile_s;
  // What lane l of DMA instruction i fetches — slot (i*64 + l) / CPR, chunk (i*64 + l) % CPR, the swizzle, whether the token's
  // grid column is window padding — does not depend on the tile: worked out once per workgroup into this table (one word per
  // (instruction of this wave, lane), read back by the lane that wrote it), so that the per-tile address of a DMA instruction is
  // a row pointer plus a table offset instead of two integer divisions and the swizzle — those were ~100 of the ~300 VALU
  // instructions per tile and wave of the global kernel (-4 %: 1087 -> 1041 us per 8 frames) and more of the windowed kernel's
  // (no change: profiles/r4_c25_* — that kernel is 70 us of prologue + 89 us of DMA / barrier skeleton + 55 us of arithmetic).
  //   bits 0-7 slot (clamped to the tile's last key), 8-11 grid row of the slot inside the tile, 12 column is padding,
  //   13 source is the constant pad page, 16-31 byte offset inside the K / V row of this head (or inside the pad page)
  constexpr int NKI = (CPR + NW - 1) / NW, NVI = (CPV + NW - 1) / NW;
  __shared__ unsigned dma_tab[NKI + NVI][NT];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  int bx, h, b;                                         // XCD-aware work order (common.h)
  flash_wg_decode(blockIdx.x, (N + QT - 1) / QT, heads, B, uh, bx, h, b);
  const int D = heads * HD;
  const long tok0 = (long)b * N;
  const int qblk = bx * QT;
  const int ql = wave * 32 + li;
  const int q = qblk + ql;
  // A wave none of whose 32 query slots is a token (14 x 14 windows: 196 tokens on 2 x 128 slots, the last wave of the second
  // workgroup) only helps staging the tiles; a half tile none of whose 32 key slots is a key (the last tile of a window holds grid
  // rows 12, 13 in slots 0 - 27) is not multiplied: 15 of the 64 (wave, half tile) units of a (window, head).  (-2 % only,
  // profiles/r4_c15_*: the kernel is not bound by its MFMA or softmax work.)
  const bool wave_live = qblk + wave * 32 < N;
  // ---- key tiles.  A tile has 64 slots holding KTV = RPT*SG keys = RPT whole rows of the SG x SG token grid (global
  //      blocks: 1 row of 64; 14x14 windows: 4 rows = 56 keys + 8 pad slots), so that the rel_w bias of a lane's 32
  //      score slots is the same for every tile (registers) and rel_h is RPT values per tile.  Pad slots and rows
  //      beyond the grid get a -inf bias (their K/V rows are clamped copies of valid keys: finite, weight 0).
  //      Everything is kept in the log2 domain: s2 = s*scale*log2e + bias*log2e, p = exp2(s2 - m2) (one v_exp_f32).
  constexpr int RPT = SG >= 64 ? 1 : 64 / SG, KTV = SG >= 64 ? 64 : RPT * SG;
  constexpr float LOG2E = 1.4426950408889634f;
  const float c2 = scale * LOG2E;
  // Window padding (FlashPad, ops.h): token (iy, ix) of window (wy, wx) lies outside the token grid when wy*SG + iy >= gh or
  // wx*SG + ix >= gw.  SAM zero-pads AFTER norm1, so such a token's qkv is the bias alone: its K / V chunks are fetched from the
  // block's bias row instead of the qkv matrix (whose pad rows nobody has written), its query is taken as zero (the
  // output rows of pad queries are never read: the proj GEMM gathers real tokens only).
  const int pw = pad.bias_row ? b % pad.nwin : 0;
  const int pwy = pw / max(pad.nwx, 1), py0 = pwy * SG, px0 = (pw - pwy * max(pad.nwx, 1)) * SG;
  auto is_pad = [&](int t) {
    const int iy = t / SG, ix = t - iy * SG;
    return pad.bias_row != nullptr && (py0 + iy >= pad.gh || px0 + ix >= pad.gw);
  };
  // DMA staging: wave w issues the K instructions i = w, w + NW, ... (CPR of them, 64 lanes x 16 B each: LDS bytes
  // [i * 1024, +1024) of the K buffer = slots (i*64 + lane) / CPR) and likewise the CPV V instructions
  {
    int n = 0;
    for (int i = wave; i < CPR; i += NW, ++n) {
      const int e = i * 64 + lane, slot = e / CPR, c = e - slot * CPR;
      const int sc = min(slot, KTV - 1), siy = sc / SG, six = sc - siy * SG;
      const unsigned off = (unsigned)((c ^ ((slot >> KSH) & KSWZ)) * 16);
      dma_tab[n][tid] = (unsigned)sc | (unsigned)siy << 8 | (pad.bias_row != nullptr && px0 + six >= pad.gw ? 1u << 12 : 0u) | off << 16;
    }
    n = NKI;
    for (int i = wave; i < CPV; i += NW, ++n) {
      const int e = i * 64 + lane, slot = e / CPV, c = e - slot * CPV;
      const int sc = min(slot, KTV - 1), siy = sc / SG, six = sc - siy * SG;
      const int cs = HD == 64 ? c ^ (((slot >> 1) & 1) << 2) : c;
      const unsigned off = (unsigned)(c < CPR ? cs * 16 : (c - CPR) * 16);
      dma_tab[n][tid] = (unsigned)sc | (unsigned)siy << 8 | (pad.bias_row != nullptr && px0 + six >= pad.gw ? 1u << 12 : 0u) |
                        (c < CPR ? 0u : 1u << 13) | off << 16;
    }
  }
  // kt0 / kh0: first key / first grid row of the tile
  auto dma_tile = [&](int kt0, int kh0, int buf) {
    const half_t* kbase = qkv + tok0 * 3 * D + D + h * HD;
    // (32-bit row offsets: one (frame | window) of qkv rows is far below 4 GB; tok0 is part of the 64-bit base)
    const unsigned stride = 3u * (unsigned)D * 2u;
    const char* kb = (const char*)kbase;
    const char* vb = (const char*)(kbase + D);
    const char* pk = (const char*)(pad.bias_row ? pad.bias_row + D + h * HD : kbase);
    const char* pv = (const char*)(pad.bias_row ? pad.bias_row + 2 * D + h * HD : kbase);
    int n = 0;
    for (int i = wave; i < CPR; i += NW, ++n) {
      const unsigned t = dma_tab[n][tid];
      const unsigned krow = min((unsigned)kt0 + (t & 255u), (unsigned)(N - 1));
      const int gr = kh0 + (int)(t >> 8 & 15u);          // grid row; rows beyond the grid (weight 0) must not touch padded rows either
      const bool pd = pad.bias_row != nullptr && ((t & 0x1000u) != 0 || gr >= SG || py0 + gr >= pad.gh);
      const char* src = pd ? pk + (t >> 16) : kb + (__umul24(krow, stride) + (t >> 16));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)((char*)&Ks[buf][0][0] + i * 1024), 16, 0, 0);
    }
    n = NKI;
    for (int i = wave; i < CPV; i += NW, ++n) {
      const unsigned t = dma_tab[n][tid];
      const unsigned vrow = min((unsigned)kt0 + (t & 255u), (unsigned)(N - 1));
      const int gr = kh0 + (int)(t >> 8 & 15u);          // grid row; rows beyond the grid (weight 0) must not touch padded rows either
      const bool pd = pad.bias_row != nullptr && ((t & 0x1000u) != 0 || gr >= SG || py0 + gr >= pad.gh);
      const char* src = pd ? pv + (t >> 16) : vb + (__umul24(vrow, stride) + (t >> 16));
      if (LROW && (t & 0x2000u)) src = (const char*)g_flash_pad + (t >> 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)((char*)&Vs[buf][0][0] + i * 1024), 16, 0, 0);
    }
  };
  // (the first K / V tile is requested after the prologue: until then the buffer holds the rel_w table)

  // ---- prologue: Q fragments, then the decomposed rel-pos tables of THIS wave's 32 queries
  //      computed with MFMA straight into LDS (fp16): G[rho][q] = <rel_pos[rho], q_vec> for all 2*SG-1 table rows, and
  //      rel_h[q][kh] = G_h[qh - kh + SG-1][q], rel_w[q][kw] = G_w[qw - kw + SG-1][q]   (App. A-3) — no HBM round trip.
  // With HD % 32 != 0 (LROW) the padded V channels are free MFMA work: channel HD is all ones (the DMA pad page), so
  // O^T[HD][q] accumulates the softmax denominator sum_k P[k][q] (of the SAME fp16-rounded P the numerator uses) with no
  // VALU adds.
  for (int i = tid; i < SG * RLD; i += NT) {
    (&relh_s[0][0])[i] = (half_t)0.f;
    (&relw_s[0][0])[i] = (half_t)0.f;
  }
  h8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (q < N && !is_pad(q)) qf[ks] = *(const h8*)(qkv + (tok0 + q) * 3 * D + h * HD + ks * 16 + hi * 8);
    else qf[ks] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();
  {
    constexpr int NR = 2 * SG - 1, NTIL = (NR + 31) / 32;
    const int qh = q / SG, qw = q - qh * SG;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const float* tab = tb == 0 ? rel_h : rel_w;
      const int qpos = tb == 0 ? qh : qw;
#pragma unroll
      for (int t = 0; t < NTIL; ++t) {
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
        const int row = 32 * t + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          h8 tf = (h8){0, 0, 0, 0, 0, 0, 0, 0};
          if (pad.rel_ops) {           // ready-made operand image (FlashPad::rel_ops): one 16-byte load
            tf = *(const h8*)(pad.rel_ops + ((((long)tb * NTIL + t) * KS + ks) * 64 + lane) * 8);
          } else if (row < NR) {
            const float4* tp = (const float4*)(tab + (long)row * HD + ks * 16 + hi * 8);
            float4 t0 = tp[0], t1 = tp[1];
            tf = (h8){(half_t)t0.x, (half_t)t0.y, (half_t)t0.z, (half_t)t0.w,
                      (half_t)t1.x, (half_t)t1.y, (half_t)t1.z, (half_t)t1.w};
          }
          g = __builtin_amdgcn_mfma_f32_32x32x16_f16(tf, qf[ks], g, 0, 0, 0);
        }
        if (q < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rho = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int kx = qpos - rho + SG - 1;
            if (kx >= 0 && kx < SG) {
              if (tb == 0) relh_s[kx][ql] = (half_t)g[r];
              else relw_s[kx][ql] = (half_t)g[r];
            }
          }
        }
      }
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  __syncthreads();                                   // rel tables of all waves are in LDS
  float relw2[32];                                   // log2e * rel_w bias of this lane's 32 slots (tile-invariant)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;      // slot
      const int kw = SG >= 64 ? j : j % SG;
      relw2[kt * 16 + r] = j < KTV ? LOG2E * (float)relw_s[kw][ql] : -INFINITY;
    }

  // transposing V reads: byte offset inside a V buffer of this lane's source chunk for k-step 0, first key quad
  int vl[DT];
  {
    const int s16 = li & 15, r4 = s16 >> 2, cc = s16 & 3, g16 = li >> 4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      vl[dt] = (4 * hi + r4) * VP * 2 + ((dt * 64 + g16 * 32 + cc * 8) ^ (HD == 64 ? ((r4 >> 1) & 1) << 6 : 0));
  }
  __syncthreads();                                   // every wave holds its rel_w values: the buffer is free for tile 0
  dma_tile(0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile 0 has landed: this wave's part; the barrier publishes all parts
  __syncthreads();
  for (int kt0 = 0, kh0 = 0, it = 0; kt0 < N; kt0 += KTV, kh0 += RPT, ++it) {
    constexpr int buf = 0;
    if (wave_live) {
    const bool half1 = RPT == 1 || kh0 + 32 / SG < SG;      // slot 32 lies in grid row kh0 + 32 / SG (wave-uniform)

    // ---- S^T = K . Q^T  (two 32-slot tiles)
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
      if (kt == 1 && !half1) continue;            // its scores stay 0 + a -inf bias: p = 0
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int krow = kt * 32 + li;
        h8 kf = *(const h8*)&Ks[buf][krow][((ks * 2 + hi) ^ ((krow >> KSH) & KSWZ)) * 8];
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], st[kt], 0, 0, 0);
      }
    }
    // ---- log2-domain scores: fma(s, scale*log2e, rel_w2[slot]) + rel_h2[row of the slot], running max
    float rh[RPT];
#pragma unroll
    for (int jr = 0; jr < RPT; ++jr) rh[jr] = kh0 + jr < SG ? LOG2E * (float)relh_s[kh0 + jr][ql] : -INFINITY;
    // (pairs of slots as 2-vectors: v_pk_fma_f32 / v_pk_add_f32 halve the VALU issue count of this VALU-bound part;
    //  with one grid row per tile (RPT == 1) the rel_h term is constant over the tile and moves into the max instead)
    float mloc = -INFINITY;
    f32x2 sv2[2][8];
    const f32x2 c2v = (f32x2){c2, c2};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        f32x2 x = (f32x2){st[kt][2 * rp], st[kt][2 * rp + 1]};
        x = x * c2v + (f32x2){relw2[kt * 16 + 2 * rp], relw2[kt * 16 + 2 * rp + 1]};
        if (RPT > 1) {
          f32x2 bb;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 2 * rp + e;
            const int j0 = kt * 32 + (r & 3) + 8 * (r >> 2);        // slot of the hi = 0 half; hi = 1 is j0 + 4
            const int ja = min(j0 / SG, RPT - 1), jb = min((j0 + 4) / SG, RPT - 1);
            bb[e] = ja == jb ? rh[ja] : (hi ? rh[jb] : rh[ja]);
          }
          x = x + bb;
        }
        sv2[kt][rp] = x;
        mloc = fmaxf(mloc, fmaxf(x[0], x[1]));
      }
    }
    if (RPT == 1) mloc += rh[0];
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);           // finite: slot 0 of every tile is a valid key
    const float msub = RPT == 1 ? m_new - rh[0] : m_new;
    const f32x2 mv = (f32x2){msub, msub};
    float lsum = 0.f;
    h8 pb[4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !half1) continue;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 e2 = sv2[kt][r >> 1] - mv;
        const float p0 = __builtin_amdgcn_exp2f(e2[0]), p1 = __builtin_amdgcn_exp2f(e2[1]);
        if (!LROW) lsum += p0 + p1;
        f32x2 pp = (f32x2){p0, p1};
        h2 ph = __builtin_convertvector(pp, h2);
        pb[kt * 2 + (r >> 3)][r & 7] = ph[0];
        pb[kt * 2 + (r >> 3)][(r & 7) + 1] = ph[1];
      }
    }
    if (!LROW) lsum += __shfl_xor(lsum, 32, 64);
    if (__any(m_new > m_run)) {                       // wave-uniform: the running max settles after a few tiles
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      m_run = m_new;
    }
    l_run += lsum;
    // ---- O^T += V^T . P^T : k-slot (hi, j) of step t is key slot 16t + 4hi + (j&3) + 8(j>>2) on BOTH operands
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= 2 && !half1) continue;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        // channel dt*32 + li of key slots 16t + 4hi + {0..3} and 16t + 8 + 4hi + {0..3}
        typedef short s4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s4 lds_s4;
        const __attribute__((address_space(3))) char* vb =
            (const __attribute__((address_space(3))) char*)&Vs[buf][0][0] + vl[dt] + 16 * t * VP * 2;
        const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)vb);
        const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(vb + 8 * VP * 2));
        const h8 vf = __builtin_bit_cast(h8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pb[t], o[dt], 0, 0, 0);
      }
    }
    }   // wave_live
    // everyone is done reading the buffer: refill it (the other resident workgroups multiply meanwhile), then wait for this
    // wave's part of the DMA; the barrier publishes all parts
    if (kt0 + KTV < N) {
      __syncthreads();
      dma_tile(kt0 + KTV, kh0 + RPT, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: out[q][h*HD + d] = O^T[d][q] / l
  if (LROW) {
    constexpr int LR = HD % 32, RL = (LR % 4) + 4 * (LR / 8), HL = (LR / 4) % 2;   // C-layout slot of O^T row HD
    l_run = __shfl(o[DT - 1][RL], li | (HL << 5), 64);
  }
  if (q < N) {
    const float inv = 1.0f / l_run;
    half_t* op = out + (tok0 + q) * D + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int d0 = dt * 32 + 8 * g + 4 * hi;
        if (d0 < HD) {
          h4 v = (h4){(half_t)(o[dt][4 * g] * inv), (half_t)(o[dt][4 * g + 1] * inv), (half_t)(o[dt][4 * g + 2] * inv),
                      (half_t)(o[dt][4 * g + 3] * inv)};
          *(h4*)(op + d0) = v;
        }
      }
    }
  }
}

// rel_h / rel_w: the block's rel_pos tables, f32 [2S-1][hd]
int vit_flash_attention_f16(const half_t* qkv, const float* relh, const float* relw, half_t* out, int B, int S,
                            int heads, int hd, hipStream_t s, FlashPad pad) {
  if (pad.bias_row && (pad.nwx <= 0 || pad.nwin <= 0 || B % pad.nwin)) return SAMPT_ERR_ARG;
  const int N = S * S;
  const float scale = 1.0f / sqrtf((float)hd);
  // work order: windows are kept whole on one XCD (common.h flash_wg_decode); global blocks keep the plain order (grouping the
  // 32 query blocks of a (frame, head) per XCD measured 0.7 % slower, profiles/r4_c16_*)
#define FL(HDv, NWv, SGv)                                                                                     \
  hipLaunchKernelGGL((k_flash_f16<HDv, NWv, SGv>), dim3(cdiv(N, NWv * 32) * heads * B), dim3(NWv * 64), 0, s, qkv, relh, relw, \
                     out, N, heads, scale, pad, B, SGv < 64 ? heads : 0)
  // (round 6, profiles/r6_c10_*: more queries per workgroup — so that a (frame, head)'s K / V tiles are staged fewer times — LOSES
  //  for the global blocks too: 6 waves / 192 queries 1607 us, 8 waves / 256 queries 1304 us against 1064 us per 8 frames with 4;
  //  7-wave windows 269 vs 210 us.  Resident workgroups, not staged bytes, are what these kernels run on.)
  if (S == 64 && hd == 80) FL(80, 4, 64);
  else if (S == 64 && hd == 64) FL(64, 4, 64);
  // (a 7-wave workgroup per (window, head) — 224 query slots for the 196 tokens instead of 2 x 128 — measured 4 % slower,
  //  profiles/r2_v7_attn_nw7.log; an 8-wave one whose two query-less waves only help staging the K / V tiles, staged once per
  //  window instead of twice: neutral, 261.2 vs 261.0 us, profiles/r3_v15_bench_attn_nw8.log — both with the register-staged
  //  tiles this kernel had before the LDS-DMA staging)
  else if (S == 14 && hd == 80) FL(80, 4, 14);
  else if (S == 14 && hd == 64) FL(64, 4, 14);
  else if (S == 16 && hd == 32) FL(32, 4, 16);   // reduced test geometry (vit_test)
  else if (S == 6 && hd == 32) FL(32, 2, 6);
  else return SAMPT_ERR_UNSUPPORTED;
#undef FL
  SAMPT_CHECK_LAUNCH("vit_flash_attention_f16");
  return SAMPT_OK;
}

}  // namespace sampt
