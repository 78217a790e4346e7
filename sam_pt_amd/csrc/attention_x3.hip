// SAM ViT attention at fp32 grade on the fp16 matrix pipe: the flash kernel of attention.hip with every product rebuilt
// from split-fp16 pieces (precision "f16x3" of the image encoder; reference arithmetic: fp32 ImageEncoderViT called at
// /root/reference/sam_pt/modeling/sam_pt.py:849, third-party segment_anything Attention).
//
//   attn = softmax(q.k^T * scale + rel_h[q, kh] + rel_w[q, kw]) . v
//
// Operands are "x3 rows" (common.h GemmP::x3): qkv [tokens][2 * 3D] halves, every block of 32 logical channels stored as
// hi(32) | lo(32) with v = hi + lo — what the x3 qkv GEMM writes; the output [tokens][2 * D] has the same format and is the
// A operand of the x3 proj GEMM.  Each MFMA product of the fp16 kernel becomes three (lo.hi + hi.lo + hi.hi, fp32
// accumulate; the dropped lo.lo term is < 2^-22 relative):
//   * S^T = K.Q^T         K tile hi / lo planes in LDS, Q hi / lo fragments in registers
//   * rel-pos tables      table rows split on the fly, results kept in LDS as fp32
//   * O^T = V^T.P^T       P = exp2(.) split in registers (ph = fp16(p), pl = fp16(p - ph)), V hi / lo planes in LDS
// Softmax statistics, rescaling and the bias are fp32 exactly as in the fp16 kernel (log2 domain, v_exp_f32).  Structure,
// key-tile geometry, LDS-DMA staging with source-side swizzles and the transposing V reads are those of k_flash_f16 — see
// the comments there; this file only notes what differs (one K / V tile buffer instead of two: see FlashX3Geom).
#include "ops.h"

namespace sampt {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ half_t g_flash_pad_x3[16] = {(half_t)1.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f,
                                        (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f,
                                        (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};

// LDS budget.  With hi and lo planes a double-buffered K / V tile pair plus two fp32 bias tables is 153 KiB for 64 x 64 tokens /
// head dim 80: ONE 4-wave workgroup per CU, one wave per SIMD.  Measured (profiles/r4_c6_attn_x3_nbuf*.log) that loses to ONE
// tile buffer with the rel_w table — dead once its values sit in registers — aliased onto it: 77 KiB, TWO workgroups per CU, each
// exposing its DMA latency once per tile while the other one multiplies (global 3779 -> 2806 us, windowed 717 -> 510 us per
// 8 frames, 50.6 -> 56.0 fps end to end in the same call).  The double-buffered variant is gone.
template <int HD, int NW, int SG>
struct FlashX3Geom {
  static constexpr int DT = (HD + 31) / 32;
  static constexpr bool LROW = DT * 32 > HD;
  static constexpr int VP = LROW ? DT * 32 : HD;
  static constexpr int QT = NW * 32, RLD = QT + 2;
  static constexpr int K_PLANE = 64 * HD * 2, V_PLANE = 64 * VP * 2;           // bytes
  static constexpr int BUF = 2 * K_PLANE + 2 * V_PLANE;                        // one buffer: K hi | K lo | V hi | V lo
  static constexpr int REL = SG * RLD * 4;
  static constexpr int TILES = BUF > REL ? BUF : REL;          // the tile buffer, which first holds the rel_w table
  static constexpr int LDS = TILES + REL;
};

template <int HD, int NW, int SG>
__global__ __launch_bounds__(NW * 64, 2) void k_flash_x3(const half_t* __restrict__ qkv, const float* __restrict__ rel_h,
                                                         const float* __restrict__ rel_w, half_t* __restrict__ out, int N,
                                                         int heads, float scale, FlashPad pad, int B, int uh) {
  typedef FlashX3Geom<HD, NW, SG> G;
  constexpr int KS = HD / 16;
  constexpr int DT = G::DT;
  constexpr int QT = G::QT;
  constexpr int NT = NW * 64;
  constexpr bool LROW = G::LROW;
  constexpr int VP = G::VP;
  constexpr int RLD = G::RLD;
  constexpr int CPR = HD / 8, CPV = VP / 8;
  constexpr int KSWZ = CPR == 8 ? 7 : (CPR == 4 ? 3 : (CPR == 10 ? 1 : 0)), KSH = CPR == 4 ? 2 : (CPR == 10 ? 3 : 1);
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  // plane p (0 = hi, 1 = lo) of buffer b
  auto Kp = [&](int b, int p) -> char* { return smem + b * G::BUF + p * G::K_PLANE; };
  auto Vp = [&](int b, int p) -> char* { return smem + b * G::BUF + 2 * G::K_PLANE + p * G::V_PLANE; };
  float* relh_s = (float*)(smem + G::TILES);            // [SG][RLD]
  float* relw_s = (float*)smem;                         // lives in the (not yet used) tile buffer

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  int bx, h, b;                                         // XCD-aware work order (common.h)
  flash_wg_decode(blockIdx.x, (N + QT - 1) / QT, heads, B, uh, bx, h, b);
  const int D = heads * HD;
  const long ldq = 6L * D;                              // halves per x3 qkv row
  const long tok0 = (long)b * N;
  const int qblk = bx * QT;
  const int ql = wave * 32 + li;
  const int q = qblk + ql;
  const bool wave_live = qblk + wave * 32 < N;   // attention.hip: token-less waves only stage, key-less half tiles are skipped
  constexpr int RPT = SG >= 64 ? 1 : 64 / SG, KTV = SG >= 64 ? 64 : RPT * SG;
  constexpr float LOG2E = 1.4426950408889634f;
  const float c2 = scale * LOG2E;

  // x3 position (halves) of logical column c0 + 8 * chunk of a qkv row: a 16-byte chunk never straddles a 32-block
  auto xcol = [](int c) -> int { return ((c >> 5) << 6) + (c & 31); };
  // window padding: pad tokens' K / V come from the block's bias row (an x3 row here), pad queries are zero (see k_flash_f16)
  const int pw = pad.bias_row ? b % pad.nwin : 0;
  const int pwy = pw / max(pad.nwx, 1), py0 = pwy * SG, px0 = (pw - pwy * max(pad.nwx, 1)) * SG;
  auto is_pad = [&](int t) {
    const int iy = t / SG, ix = t - iy * SG;
    return pad.bias_row != nullptr && (py0 + iy >= pad.gh || px0 + ix >= pad.gw);
  };
  auto dma_tile = [&](int kt0, int buf) {
    const half_t* rbase = qkv + tok0 * ldq;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      for (int i = wave; i < CPR; i += NW) {
        const int e = i * 64 + lane, slot = e / CPR, c = e - slot * CPR;
        const int krow = min(kt0 + min(slot, KTV - 1), N - 1);
        const half_t* rowp = is_pad(krow) ? pad.bias_row : rbase + (long)krow * ldq;
        const half_t* src = rowp + xcol(D + h * HD + ((c ^ ((slot >> KSH) & KSWZ)) * 8)) + pl * 32;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(Kp(buf, pl) + i * 1024), 16, 0, 0);
      }
      for (int i = wave; i < CPV; i += NW) {
        const int e = i * 64 + lane, slot = e / CPV, c = e - slot * CPV;
        const int vrow = min(kt0 + min(slot, KTV - 1), N - 1);
        const int cs = HD == 64 ? c ^ (((slot >> 1) & 1) << 2) : c;
        // pad channels: hi plane = the page whose first half is 1.0 (channel HD of V is all ones), lo plane = zeros
        const half_t* rowp = is_pad(vrow) ? pad.bias_row : rbase + (long)vrow * ldq;
        const half_t* src = c < CPR ? rowp + xcol(2 * D + h * HD + cs * 8) + pl * 32
                                    : g_flash_pad_x3 + (pl == 0 ? (c - CPR) * 8 : 8);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(Vp(buf, pl) + i * 1024), 16, 0, 0);
      }
    }
  };
  // (tile 0 is requested only after the prologue: until then the buffer holds the rel_w table)

  for (int i = tid; i < SG * RLD; i += NT) relh_s[i] = 0.f, relw_s[i] = 0.f;
  h8 qh[KS], qlo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (q < N && !is_pad(q)) {
      const half_t* qp = qkv + (tok0 + q) * ldq + xcol(h * HD + ks * 16 + hi * 8);
      qh[ks] = *(const h8*)qp;
      qlo[ks] = *(const h8*)(qp + 32);
    } else {
      qh[ks] = qlo[ks] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  __syncthreads();
  {
    constexpr int NR = 2 * SG - 1, NTIL = (NR + 31) / 32;
    const int qhh = q / SG, qw = q - qhh * SG;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const float* tab = tb == 0 ? rel_h : rel_w;
      const int qpos = tb == 0 ? qhh : qw;
      float* dst = tb == 0 ? relh_s : relw_s;
#pragma unroll
      for (int t = 0; t < NTIL; ++t) {
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
        const int row = 32 * t + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          h8 th = (h8){0, 0, 0, 0, 0, 0, 0, 0}, tl = th;
          if (row < NR) {
            const float4* tp = (const float4*)(tab + (long)row * HD + ks * 16 + hi * 8);
            const float4 t0 = tp[0], t1 = tp[1];
            const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              half_t a, bb;
              split_f16(tv[e], a, bb);
              th[e] = a, tl[e] = bb;
            }
          }
          g = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl, qh[ks], g, 0, 0, 0);
          g = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, qlo[ks], g, 0, 0, 0);
          g = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, qh[ks], g, 0, 0, 0);
        }
        if (q < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rho = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int kx = qpos - rho + SG - 1;
            if (kx >= 0 && kx < SG) dst[kx * RLD + ql] = g[r];
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

  __syncthreads();
  float relw2[32];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int kw = SG >= 64 ? j : j % SG;
      relw2[kt * 16 + r] = j < KTV ? LOG2E * relw_s[kw * RLD + ql] : -INFINITY;
    }

  int vl[DT];
  {
    const int s16 = li & 15, r4 = s16 >> 2, cc = s16 & 3, g16 = li >> 4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      vl[dt] = (4 * hi + r4) * VP * 2 + ((dt * 64 + g16 * 32 + cc * 8) ^ (HD == 64 ? ((r4 >> 1) & 1) << 6 : 0));
  }
  __syncthreads();                        // every wave has its rel_w values in registers: the buffer is free for tile 0
  dma_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt0 = 0, kh0 = 0, it = 0; kt0 < N; kt0 += KTV, kh0 += RPT, ++it) {
    constexpr int buf = 0;
    if (wave_live) {
    // (the key-less half tile is skipped only where that keeps the register count: head dim 64 would drop from 3 to 2 waves per SIMD)
    const bool half1 = RPT == 1 || HD == 64 || kh0 + 32 / SG < SG;

    // ---- S^T = K . Q^T, three products per k-step
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
      if (kt == 1 && !half1) continue;            // its scores stay 0 + a -inf bias: p = 0
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int krow = kt * 32 + li;
        const int off = (krow * HD + ((ks * 2 + hi) ^ ((krow >> KSH) & KSWZ)) * 8) * 2;
        const h8 kfh = *(const h8*)(Kp(buf, 0) + off);
        const h8 kfl = *(const h8*)(Kp(buf, 1) + off);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl, qh[ks], st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, qlo[ks], st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, qh[ks], st[kt], 0, 0, 0);
      }
    }
    float rh[RPT];
#pragma unroll
    for (int jr = 0; jr < RPT; ++jr) rh[jr] = kh0 + jr < SG ? LOG2E * relh_s[(kh0 + jr) * RLD + ql] : -INFINITY;
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
            const int j0 = kt * 32 + (r & 3) + 8 * (r >> 2);
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
    const float m_new = fmaxf(m_run, mloc);
    // P is kept scaled by 2^PSH (p' = exp2(s2 - m + PSH) <= 2^15 < 65504): the lo piece of its split, p' * 2^-12, then stays a
    // NORMAL fp16 number down to p = 2^-17 of the row maximum instead of falling into the subnormals at p < 2^-3; the factor
    // is common to O and to the softmax denominator and cancels in O / l.
    constexpr float PSH = 15.f;
    const float msub = (RPT == 1 ? m_new - rh[0] : m_new) - PSH;
    const f32x2 mv = (f32x2){msub, msub};
    float lsum = 0.f;
    h8 pbh[4], pbl[4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !half1) continue;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 e2 = sv2[kt][r >> 1] - mv;
        const float p0 = __builtin_amdgcn_exp2f(e2[0]), p1 = __builtin_amdgcn_exp2f(e2[1]);
        if (!LROW) lsum += p0 + p1;
        const f32x2 pp = (f32x2){p0, p1};
        const h2 ph = __builtin_convertvector(pp, h2);
        const f32x2 pr = pp - __builtin_convertvector(ph, f32x2);
        const h2 pl = __builtin_convertvector(pr, h2);
        pbh[kt * 2 + (r >> 3)][r & 7] = ph[0];
        pbh[kt * 2 + (r >> 3)][(r & 7) + 1] = ph[1];
        pbl[kt * 2 + (r >> 3)][r & 7] = pl[0];
        pbl[kt * 2 + (r >> 3)][(r & 7) + 1] = pl[1];
      }
    }
    if (!LROW) lsum += __shfl_xor(lsum, 32, 64);
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      m_run = m_new;
    }
    l_run += lsum;
    // ---- O^T += V^T . P^T, three products per k-step (the ones channel of V hi sums ph + pl: the softmax denominator)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= 2 && !half1) continue;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        typedef short s4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s4 lds_s4;
        h8 vf[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const __attribute__((address_space(3))) char* vb =
              (const __attribute__((address_space(3))) char*)Vp(buf, pl) + vl[dt] + 16 * t * VP * 2;
          const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)vb);
          const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(vb + 8 * VP * 2));
          vf[pl] = __builtin_bit_cast(h8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1], pbh[t], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], pbl[t], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], pbh[t], o[dt], 0, 0, 0);
      }
    }
    }   // wave_live
    if (kt0 + KTV < N) {
      __syncthreads();                    // everyone is done reading the only buffer
      dma_tile(kt0 + KTV, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: out x3 row q, logical columns h*HD + d
  if (LROW) {
    constexpr int LR = HD % 32, RL = (LR % 4) + 4 * (LR / 8), HL = (LR / 4) % 2;
    l_run = __shfl(o[DT - 1][RL], li | (HL << 5), 64);
  }
  if (q < N) {
    const float inv = 1.0f / l_run;
    half_t* op = out + (tok0 + q) * 2 * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * hi;
        if (d0 < HD) {
          h4 vh, vlo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            half_t a, bb;
            split_f16(o[dt][4 * g + e] * inv, a, bb);
            vh[e] = a, vlo[e] = bb;
          }
          half_t* p = op + xcol(h * HD + d0);
          *(h4*)p = vh;
          *(h4*)(p + 32) = vlo;
        }
      }
    }
  }
}

// qkv / out: x3 rows (see the file header); rel_h / rel_w: the block's rel_pos tables, f32 [2S-1][hd]
int vit_flash_attention_x3(const half_t* qkv, const float* relh, const float* relw, half_t* out, int B, int S, int heads,
                           int hd, hipStream_t s, FlashPad pad) {
  if (pad.bias_row && (pad.nwx <= 0 || pad.nwin <= 0 || B % pad.nwin)) return SAMPT_ERR_ARG;
  const int N = S * S;
  const float scale = 1.0f / sqrtf((float)hd);
  if ((heads * hd) % 32) return SAMPT_ERR_UNSUPPORTED;
#define FLX(HDv, NWv, SGv)                                                                                              \
  do {                                                                                                                  \
    typedef FlashX3Geom<HDv, NWv, SGv> G;                                                                               \
    static bool raised = false;                                                                                         \
    auto kern = k_flash_x3<HDv, NWv, SGv>;                                                                              \
    if (!raised) {                                                                                                      \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)     \
        return SAMPT_ERR_HIP;                                                                                           \
      raised = true;                                                                                                    \
    }                                                                                                                   \
    hipLaunchKernelGGL(kern, dim3(cdiv(N, NWv * 32) * heads * B), dim3(NWv * 64), G::LDS, s, qkv, relh, relw, out, N,   \
                       heads, scale, pad, B, SGv < 64 ? heads : 0);   /* work order: as attention.hip */                \
  } while (0)
  if (S == 64 && hd == 80) FLX(80, 4, 64);
  else if (S == 64 && hd == 64) FLX(64, 4, 64);
  else if (S == 14 && hd == 80) FLX(80, 4, 14);
  else if (S == 14 && hd == 64) FLX(64, 4, 14);
  else if (S == 16 && hd == 32) FLX(32, 4, 16);   // reduced test geometry (vit_test)
  else if (S == 6 && hd == 32) FLX(32, 2, 6);
  else return SAMPT_ERR_UNSUPPORTED;
#undef FLX
  SAMPT_CHECK_LAUNCH("vit_flash_attention_x3");
  return SAMPT_OK;
}

}  // namespace sampt
