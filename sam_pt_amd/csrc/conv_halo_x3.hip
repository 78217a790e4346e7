// 3 x 3 (stride 1, pad 1) split-fp16 convolution over pre-split NHWC planes with the input tile's HALO staged once.
//
// Why: k_conv_f16x3_dma (conv_f16x3.hip) is an implicit GEMM whose K loop walks (tap, channel chunk) and fetches the 128 A rows of
// its tile again for every tap: 9 x the input bytes through the CU, plus the whole weight tile per 128 pixels.  A CU moves
// 35 - 38 GB/s through LDS-DMA on these launches (L2 hits included) and that, not the matrix pipe (22 - 35 % busy), is what they
// run at: 64 -> 64 at 288 x 512 x 8 frames 419 us for 442 KB per workgroup, the 416 -> 256 convolution 1.92 ms for 3.8 MB per
// workgroup (profiles/r6_c7_clip_kernels_by_grid*).  Here a workgroup owns a 16 x 16 pixel tile:
//   * per 32-channel chunk the 18 x 18 halo is staged ONCE (both planes, 42 KB, double-buffered) and all nine taps read it at
//     shifted rows — 9 x fewer input bytes;
//   * the weights of a (chunk, tap) are a 4-slot LDS ring of BN x 64 B x 2 planes, three stages in flight, ONE barrier per stage;
//     256 pixels per workgroup halve the weight bytes per pixel;
//   * a wave owns 4 image rows of the tile (4 fragments of 16 pixels) x all BN output channels: 12 BN / 16 MFMAs per stage against
//     4 + BN / 8 ds_read_b128 — the matrix pipe, not LDS or DMA issue, is the longest pole of a stage.
// Same arithmetic as the kernel it replaces (hi.lo + lo.hi + hi.hi, fp32 accumulate, 2^-8, bias), same swizzle convention (an
// LDS-DMA image is lane-linear, so the lane that fills chunk position q of LDS row r fetches source chunk q ^ F[(r >> 2) & 3],
// F = {0, 3, 2, 1}; fragment reads apply the same XOR), taps outside the image read a page of zeros.
#include <type_traits>

#include "ops.h"

namespace sampt {

namespace {
typedef __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(3))) void lds_void;
template <int T, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (T < N) {
    f(std::integral_constant<int, T>{});
    static_for<T + 1, N>(f);
  }
}
__device__ __attribute__((aligned(128))) half_t g_halo_zero_page[64];
constexpr int HT = 16, HW = 18, NPX = HW * HW;          // output tile edge, halo edge, halo pixels (324)
constexpr int APIECES = 21;                             // 16-row DMA pieces per plane (336 rows >= 324)
constexpr int APL = APIECES * 1024;                     // bytes of one A plane
constexpr int ABUF = 2 * APL;                           // one halo chunk: hi | lo
}  // namespace

// A2: the halo chunk is double-buffered (the next chunk arrives under this chunk's nine taps; 150 KB of LDS at BN = 128, one
// workgroup per CU).  A2 = false (BN = 64): ONE halo buffer, 76 KB, TWO workgroups per CU — with two chunks per tile (the 64-channel
// layers) a workgroup's fixed costs (first DMA round trip, epilogue stores: ~14 of its ~20 us) weigh more than the one exposed
// chunk boundary, and the second workgroup's waves fill the matrix pipe meanwhile (356 -> 280 us, profiles/r6_c14_*).
// NWV: waves per workgroup.  4: a wave owns 4 image rows x all BN channels.  8 (BN >= 96, where registers and LDS allow only one
// workgroup per CU): waves w and w + 4 share the 4 rows and split the channels, so every SIMD holds TWO waves — one multiplies while
// the other issues its LDS-DMA (60 - 185 issue cycles per instruction) or waits for LDS.
template <int BN, bool A2, int NWV>
__global__ __launch_bounds__(NWV * 64, (A2 && NWV == 4) ? 1 : 2) void k_conv3x3_halo_x3(GemmP p, int ntx, int nty, int xcd_order) {
  constexpr int FN = BN / 16;                  // output-channel fragments of the tile
  constexpr int FNW = FN / (NWV / 4);          // ... of one wave
  constexpr int NBP = 2 * FN;                  // W DMA pieces per stage: [hi rows | lo rows], 16 rows each
  constexpr int NW = (NBP + NWV - 1) / NWV;    // W DMA instructions per wave and stage (padded with dummies)
  constexpr int KA = (APIECES + NWV - 1) / NWV;   // halo pieces per wave and plane
  constexpr int NA = 2 * KA;                   // A DMA instructions per wave and chunk (padded with dummies)
  constexpr int WSTG = BN * 64 * 2;            // bytes of one weight stage (hi rows | lo rows)
  constexpr int NAB = A2 ? 2 : 1;
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // [A buf 0 (| A buf 1) | dummy 1 KB | W ring (4 stages)]
  char* const a_lds = lds;
  char* const dummy_lds = lds + NAB * ABUF;
  char* const w_lds = lds + NAB * ABUF + 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pw = wave & 3, ch = wave >> 2;     // pixel-row group, channel half
  const int lr = lane & 15, lq = lane >> 4;
  // ---- tile number b = ((img * nty + ty) * ntx + tx) * ntn + tn.  Workgroups go round robin over the 8 XCDs (blockIdx & 7), each
  // with its own L2: XCD x takes the CONTIGUOUS tile range [x * per, (x + 1) * per), so that the tiles that share halo rows and
  // columns (and the column tiles that share the whole halo) meet in one L2 (g_halo_xcd = 0: plain order, for A / B runs)
  const int ntn = (p.N + BN - 1) / BN;
  const int ntiles_all = p.M / (p.OH * p.OW) * nty * ntx * ntn;
  int b = blockIdx.x;
  if (xcd_order) {
    const int per = (ntiles_all + 7) >> 3;
    b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || b >= ntiles_all) return;
  } else if (b >= ntiles_all) {
    return;
  }
  const int tn = b % ntn;  b /= ntn;
  const int tx = b % ntx;  b /= ntx;
  const int ty = b % nty;
  const int img = b / nty;
  const int x0 = tx * HT, y0 = ty * HT, n0 = tn * BN;
  const int H = p.cH, W = p.cW, C = p.cC;
  const int nchunk = C / 32, nstage = nchunk * 9;
  const char* __restrict__ Ah = (const char*)p.A;
  const char* __restrict__ Al = (const char*)p.A_lo;
  const char* __restrict__ Wh = (const char*)p.W;
  const char* __restrict__ Wl = (const char*)p.W_lo;
  const char* zero = (const char*)g_halo_zero_page;

  // ---- DMA roles.  Lane l of an instruction fills LDS row (l >> 2), chunk position (l & 3) of a 16-row piece.
  const int prow = lane >> 2;
  const int fsw = (4 - (lane >> 4)) & 3;                                 // F[(row >> 2) & 3] with (row >> 2) & 3 == lane >> 4
  const int csrc = ((lane & 3) ^ fsw) * 16;                              // byte offset of this lane's source chunk in a 64-B slab row
  // A: this wave fills halo pieces wave + NWV k (pieces >= 21 are dummies): byte offset of the halo pixel's channel 0 in a plane,
  // or -1 when the pixel lies outside the image (zero page) / the piece does not exist
  long a_off[KA];
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    const int piece = wave + NWV * k, hl = piece * 16 + prow;
    const int hy = hl / HW, hx = hl - hy * HW;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = piece < APIECES && hl < NPX && y >= 0 && y < H && x >= 0 && x < W;
    a_off[k] = ok ? (((long)img * H + y) * W + x) * C * 2 + csrc : -1;
  }
  // W: pieces q = wave + NWV j over [hi rows | lo rows] (q >= NBP: dummy); row n of the weight matrix is [9 taps][C] halves
  long w_off[NW];
  bool w_lo[NW];
  int w_dst[NW];                                                         // byte offset inside a stage, or -1: dummy
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int q = wave + NWV * j;
    w_lo[j] = q >= FN;
    const int piece = w_lo[j] ? q - FN : q;
    int n = n0 + (piece < FN ? piece : 0) * 16 + prow;
    n = n < p.N ? n : p.N - 1;
    w_off[j] = (long)n * p.ldw * 2 + csrc;
    w_dst[j] = q < NBP ? (w_lo[j] ? BN * 64 : 0) + piece * 1024 : -1;
  }
  auto stage_a = [&](int c) {                                            // halo chunk c (clamped: a dummy re-issue past the end)
    const int cc = c < nchunk ? c : nchunk - 1;
    char* dst = a_lds + (A2 ? (c & 1) : 0) * ABUF;
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int piece = wave + NWV * k;                                  // uniform
      const bool real = piece < APIECES;
      const char* sh = a_off[k] >= 0 ? Ah + a_off[k] + cc * 64 : zero + csrc;
      const char* sl = a_off[k] >= 0 ? Al + a_off[k] + cc * 64 : zero + csrc;
      char* d = real ? dst + piece * 1024 : dummy_lds;
      __builtin_amdgcn_global_load_lds((glb_void*)sh, (lds_void*)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)sl, (lds_void*)(real ? d + APL : dummy_lds), 16, 0, 0);
    }
  };
  auto stage_w = [&](int s) {                                            // weights of stage s = chunk s / 9, tap s % 9
    const int ss = s < nstage ? s : nstage - 1;
    const int c = ss / 9, t = ss - c * 9;
    const long koff = ((long)t * C + c * 32) * 2;
    char* dst = w_lds + (s & 3) * WSTG;
#pragma unroll
    for (int j = 0; j < NW; ++j)
      __builtin_amdgcn_global_load_lds((glb_void*)((w_lo[j] ? Wl : Wh) + w_off[j] + koff),
                                       (lds_void*)(w_dst[j] >= 0 ? dst + w_dst[j] : dummy_lds), 16, 0, 0);
  };

  f32x4 acc[4][FNW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FNW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage_a(0);
  stage_w(0);
  stage_w(1);
  stage_w(2);
  const int wsw = ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) * 16);             // swizzled chunk of weight row j * 16 + lr
  for (int c = 0; c < nchunk; ++c) {
    const char* abuf = a_lds + (A2 ? (c & 1) : 0) * ABUF;
    static_for<0, 9>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int ky = t / 3, kx = t % 3;
      const int s = c * 9 + t;
      // this wave's share of W(s) — and, at t = 0, of the halo chunk issued a whole chunk ago — has landed.  Younger than W(s): the
      // NW instructions each of W(s + 1), W(s + 2), and for t = 1 .. 3 the NA of halo chunk c + 1 (issued at t = 0 behind W(s + 3))
      // (A2 = false: the halo chunk is the YOUNGEST thing in flight at t = 0 of every chunk but the first — everything must land)
      if (A2 && t >= 1 && t <= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW + NA) : "memory");
      else if (!A2 && t == 0 && c > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW) : "memory");
      // ... and NO LDS read of this wave may still be in flight when it signals the barrier: hipcc moves a stage's last MFMAs — and
      // with them the lgkmcnt wait of the last two weight reads — below the barrier, and the slot they read is the one the first
      // wave through the barrier re-stages at once.  The weights are hot in the vector L1, so that DMA can land within the latency of
      // a queued ds_read: 1 launch in ~15 of the 64-column variant (two workgroups per CU) had one wave's last column fragment off by
      // one stage's lo plane (tools/probes/conv_stats_determinism.py, profiles/r6_c29 - c31).
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                      // publishes all shares; everybody is done with stage s - 1
      asm volatile("" ::: "memory");
      stage_w(s + 3);                                                    // into the slot of stage s - 1
      if (A2 && t == 0) stage_a(c + 1);                                  // into the buffer of chunk c - 1
      const char* wslot = w_lds + (s & 3) * WSTG;
      h8 ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (4 * pw + i + ky) * HW + lr + kx;                  // halo row of output pixel (4 pw + i, lr) for this tap
        const int off = r * 64 + ((lq ^ ((4 - ((r >> 2) & 3)) & 3)) * 16);
        ah[i] = *(const h8*)(abuf + off);
        al[i] = *(const h8*)(abuf + APL + off);
      }
#pragma unroll
      for (int j = 0; j < FNW; ++j) {
        const int jg = ch * FNW + j;
        const h8 bh = *(const h8*)(wslot + (jg * 16 + lr) * 64 + wsw);
        const h8 bl = *(const h8*)(wslot + BN * 64 + (jg * 16 + lr) * 64 + wsw);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], acc[i][j], 0, 0, 0);
      }
      if (!A2 && t == 8 && c + 1 < nchunk) {                             // single halo buffer: refill it once nobody reads it any more
        asm volatile("" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_a(c + 1);
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the dummy tail stages must land before the LDS is handed on

  // ---- epilogue: lane (lr, lq) reg r = channel n0 + 16 jg + 4 lq + r of pixel (y0 + 4 pw + i, x0 + lr).  The bias is loaded
  // once, unconditionally (clamped columns): a load under the per-fragment branches would wait for every earlier store.
  float4 bb[FNW];
#pragma unroll
  for (int j = 0; j < FNW; ++j) {
    const int col = n0 + (ch * FNW + j) * 16 + lq * 4;
    bb[j] = p.bias ? *(const float4*)(p.bias + (col < p.N ? col : 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < FNW; ++j)     // "use" every bias register HERE: hipcc otherwise waits vmcnt(0) at each first use inside the
    asm volatile("" ::"v"(bb[j].x), "v"(bb[j].y), "v"(bb[j].z), "v"(bb[j].w));   // store branches — i.e. for the previous store
  const int x = x0 + lr;
  float* orow[4];
  bool pix_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = y0 + 4 * pw + i;
    pix_ok[i] = y < H && x < W;
    orow[i] = (float*)p.C + (((long)img * H + (pix_ok[i] ? y : 0)) * W + (pix_ok[i] ? x : 0)) * p.ldc;
  }
  // p.in_part: the InstanceNorm that follows wants sum(v), sum(v^2) per (image, channel) — this tile's share is summed here, from
  // the registers that are being stored, instead of in a pass of its own over the map.  A lane adds up its 4 rows in fp32 (three
  // roundings) and parks the 8 sums in LDS; thread c then adds the 64 (row group, pixel column) terms of channel c in fp64, as
  // k_instnorm_final does across tiles — nothing in between rounds to fp32, so |mean| >> std costs no digits.
  float* const red1 = (float*)lds;              // [row group 4][lr 16][BN] sums; the staging buffers are dead (barrier below)
  float* const red2 = red1 + 64 * BN;           // ... of squares
  if (p.in_part) __syncthreads();
#pragma unroll
  for (int j = 0; j < FNW; ++j) {
    const int jg = ch * FNW + j, col = n0 + jg * 16 + lq * 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = make_float4(acc[i][j][0] * p.alpha + bb[j].x, acc[i][j][1] * p.alpha + bb[j].y,
                                   acc[i][j][2] * p.alpha + bb[j].z, acc[i][j][3] * p.alpha + bb[j].w);
      if (pix_ok[i] && col < p.N) *(float4*)(orow[i] + col) = v;
      if (p.in_part && pix_ok[i]) {
        s1.x += v.x, s1.y += v.y, s1.z += v.z, s1.w += v.w;
        s2.x += v.x * v.x, s2.y += v.y * v.y, s2.z += v.z * v.z, s2.w += v.w * v.w;
      }
    }
    if (p.in_part) {
      *(float4*)(red1 + (pw * 16 + lr) * BN + jg * 16 + lq * 4) = s1;
      *(float4*)(red2 + (pw * 16 + lr) * BN + jg * 16 + lq * 4) = s2;
    }
  }
  if (p.in_part) {
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll 8
      for (int g = 0; g < 64; ++g) t1 += (double)red1[g * BN + tid], t2 += (double)red2[g * BN + tid];
      double* o = p.in_part + ((((long)img * nty + ty) * ntx + tx) * p.N + n0 + tid) * 2;
      o[0] = t1, o[1] = t2;
    }
  }
}

int g_conv_halo = 1;     // sampt_conv_set_halo: 0 = the 3 x 3 stride-1 launches go back to k_conv_f16x3_dma (A / B); 2 = 4-wave
                         // workgroups for every tile width (the first version of this kernel)
int g_halo_dbg = 0;      // (debug: 6 = the 64-column tile on the double-buffered schedule, 7 = one workgroup per CU, 8 = plain tile order)
int g_conv_in_stats = 3; // bit 0: the halo convolutions, bit 1: the stem sum the following InstanceNorm's statistics (sampt_conv_set_halo(3 / 4 / 5))

bool conv3x3_halo_eligible(const GemmP& p) {
  return p.conv && p.A_lo && p.KH == 3 && p.KW == 3 && p.cstride == 1 && p.cpad == 1 && p.cpadw < 0 && p.cC % 32 == 0 &&
         p.K == 9 * p.cC && p.ldw == p.K && p.OH == p.cH && p.OW == p.cW && p.act == ACT_NONE && !p.res && !p.shuf_g &&
         p.N % 4 == 0 && p.ldc % 4 == 0 && p.M == (p.M / (p.OH * p.OW)) * p.OH * p.OW;
}

// tiles per image = chunks of GemmP::in_part
int conv3x3_halo_tiles(const GemmP& p) { return cdiv(p.cW, HT) * cdiv(p.cH, HT); }

int conv3x3_halo_x3(const GemmP& p, hipStream_t s) {
  const int nimg = p.M / (p.OH * p.OW);
  const int ntx = cdiv(p.cW, HT), nty = cdiv(p.cH, HT);
  const int BN = p.N <= 64 ? 64 : (p.N <= 96 ? 96 : 128);
  const int ntn = cdiv(p.N, BN);
  const long ntiles = (long)nimg * nty * ntx * ntn;
  dim3 grid((unsigned)(((ntiles + 7) / 8) * 8));                      // (8 x per: see the XCD order in the kernel)
#define HALO(BNv, NWVv, A2x)                                                                                             \
  do {                                                                                                                   \
    constexpr bool A2v = A2x;                                                                                            \
    constexpr int LDSB = (A2v ? 2 : 1) * ABUF + 1024 + 4 * (BNv * 64 * 2);                                               \
    static bool raised = false;                                                                                          \
    auto kern = k_conv3x3_halo_x3<BNv, A2v, NWVv>;                                                                       \
    if (!raised) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024 > LDSB ? 100 * 1024 : LDSB) != hipSuccess)        \
        return SAMPT_ERR_HIP;                                                                                            \
      raised = true;                                                                                                     \
    }                                                                                                                    \
    hipLaunchKernelGGL(kern, grid, dim3(NWVv * 64), (g_halo_dbg == 7 && LDSB < 100 * 1024) ? 100 * 1024 : LDSB, s, p, ntx, nty, g_halo_dbg == 8 ? 0 : 1);  \
  } while (0)
  const bool w8 = g_conv_halo != 2;
  if (BN == 64) { if (g_halo_dbg == 6) HALO(64, 4, true); else HALO(64, 4, false); }
  else if (BN == 96) { if (w8) HALO(96, 8, true); else HALO(96, 4, true); }
  else { if (w8) HALO(128, 8, true); else HALO(128, 4, true); }
#undef HALO
  SAMPT_CHECK_LAUNCH("conv3x3_halo_x3");
  return SAMPT_OK;
}

}  // namespace sampt
