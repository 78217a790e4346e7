// The tracker encoder's stem — Conv2d(3, 64, kernel 7, stride 2, padding 3) over the normalised frames (pips.py:200 BasicEncoder.conv1) —
// as 3-term split-fp16 MFMA products, with the statistics of the InstanceNorm that follows it summed in the epilogue.
//
// Why a kernel of its own: Cin = 3 (4 with the zero channel of the NHWC4 frames) fits no 32-channel K slab, so the stem ran as an
// implicit GEMM on the exact-fp32 MFMA path: 538 us per 8 frames of 576 x 1024 for 22 GFLOP, a write of 302 MB and a read of 75
// (profiles/r6_c19_*).  Here one K slab is one KERNEL ROW: 7 taps x 4 channels = 28 k, padded to 32 with a zero-weight eighth tap:
//   * a workgroup owns a 16 x 16 tile of output pixels; the 37 x 38 input pixels under it are read once, split into fp16 hi / lo
//     (x = hi + lo, 22 bits) and kept pixel-major in LDS (8 B per pixel and plane), so that the A operand of output pixel (oy, ox),
//     kernel row ky, k-chunk lq is the 16 contiguous bytes of input pixels 2 ox + 2 lq, + 1 in patch row 2 oy + ky;
//   * the weights (64 x 7 x 32 halves x 2 planes = 56 KB, scaled by 2^8 like every split-fp16 weight) are split once per workgroup
//     into MFMA operand images in LDS; workgroups are persistent (two per CU) so that this happens 512 times, not 4608;
//   * a wave owns 4 rows of the tile x all 64 channels: per kernel row 8 + 8 ds_read_b128 against 48 MFMAs;
//   * same epilogue as conv_halo_x3.hip: alpha, bias, 16-byte stores, per-tile InstanceNorm partial sums (GemmP::in_part layout).
#include "ops.h"

namespace sampt {

namespace {
constexpr int ST = 16;                 // output tile edge
constexpr int PR = 2 * ST + 5;         // patch rows (37)
constexpr int PC = 2 * ST + 6;         // patch columns incl. the pad tap's (38): a row is 304 B, a multiple of 16
constexpr int PLANE = PR * PC * 8;     // bytes of one plane of the patch (11 248)
constexpr int WIMG = 7 * 4 * 2 * 1024; // weight images (ky, j, plane)

__global__ __launch_bounds__(256, 2) void k_stem7x7s2_x3(const float4* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         double* __restrict__ in_part, int nimg, int H, int W, int OH, int OW, int ntx,
                                                         int nty) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // [W images 56 KB | patch hi | patch lo]
  char* const w_lds = lds;
  char* const p_hi = lds + WIMG;
  char* const p_lo = p_hi + ((PLANE + 15) & ~15);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const float wscale = (float)(1 << F16X3_WSHIFT), alpha = 1.0f / wscale;

  // ---- weights, once: image (ky, j, plane), lane (lr, lq) = W[16 j + lr][ky][kx = 2 lq + (e >> 2)][c = e & 3], e = 0 .. 7; kx = 7 is 0
  for (int im = wave; im < 28; im += 4) {
    const int ky = im >> 2, j = im & 3;
    const float* wr = w + ((long)(16 * j + lr) * 49 + ky * 7) * 4;
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kx = 2 * lq + (e >> 2);
      const float v = kx < 7 ? wr[kx * 4 + (e & 3)] * wscale : 0.f;
      half_t a, b;
      split_f16(v, a, b);
      hi[e] = a, lo[e] = b;
    }
    *(h8*)(w_lds + (im * 2) * 1024 + lane * 16) = hi;
    *(h8*)(w_lds + (im * 2 + 1) * 1024 + lane * 16) = lo;
  }
  float4 bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bb[j] = bias ? *(const float4*)(bias + 16 * j + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);

  const int ntiles = nimg * nty * ntx;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % ntx, ty = (tile / ntx) % nty, img = tile / (ntx * nty);
    const int ox0 = tx * ST, oy0 = ty * ST;
    __syncthreads();                                                // the previous tile's patch (and the sums parked in it) are dead (and the weights landed)
    // ---- the input patch: rows 2 oy0 - 3 .. + 36, columns 2 ox0 - 3 .. + 37, zero outside the frame
    for (int idx = tid; idx < PR * PC; idx += 256) {
      const int r = idx / PC, c = idx - r * PC;
      const int gy = 2 * oy0 - 3 + r, gx = 2 * ox0 - 3 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[((long)img * H + gy) * W + gx];
      const float vv[4] = {v.x, v.y, v.z, v.w};
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        half_t a, b;
        split_f16(vv[e], a, b);
        hi[e] = a, lo[e] = b;
      }
      *(h4*)(p_hi + idx * 8) = hi;
      *(h4*)(p_lo + idx * 8) = lo;
    }
    __syncthreads();

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      h8 ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = ((2 * (4 * wave + i) + ky) * PC + 2 * lr + 2 * lq) * 8;
        ah[i] = *(const h8*)(p_hi + off);
        al[i] = *(const h8*)(p_lo + off);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const h8 bh = *(const h8*)(w_lds + ((ky * 4 + j) * 2) * 1024 + lane * 16);
        const h8 bl = *(const h8*)(w_lds + ((ky * 4 + j) * 2 + 1) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], acc[i][j], 0, 0, 0);
      }
    }

    // ---- epilogue: lane (lr, lq) reg r = channel 16 j + 4 lq + r of pixel (oy0 + 4 wave + i, ox0 + lr); InstanceNorm partial sums as
    // in conv_halo_x3.hip
    const int ox = ox0 + lr;
    float* orow[4];
    bool pix_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = oy0 + 4 * wave + i;
      pix_ok[i] = oy < OH && ox < OW;
      orow[i] = y + (((long)img * OH + (pix_ok[i] ? oy : 0)) * OW + (pix_ok[i] ? ox : 0)) * 64;
    }
    // (statistics: a lane's 4-row fp32 sums go to LDS, thread c adds channel c's 64 terms in fp64 — two passes of 32 channels over
    //  the dead patch, 16 KB each)
    float* const red1 = (float*)p_hi;             // [row group 4][lr 16][32]
    float* const red2 = red1 + 64 * 32;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (in_part) __syncthreads();               // everybody is done with the patch / with the previous pass
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = half * 2 + jj;
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = make_float4(acc[i][j][0] * alpha + bb[j].x, acc[i][j][1] * alpha + bb[j].y, acc[i][j][2] * alpha + bb[j].z,
                                       acc[i][j][3] * alpha + bb[j].w);
          if (pix_ok[i]) {
            *(float4*)(orow[i] + 16 * j + 4 * lq) = v;
            s1.x += v.x, s1.y += v.y, s1.z += v.z, s1.w += v.w;
            s2.x += v.x * v.x, s2.y += v.y * v.y, s2.z += v.z * v.z, s2.w += v.w * v.w;
          }
        }
        if (in_part) {
          *(float4*)(red1 + (wave * 16 + lr) * 32 + jj * 16 + lq * 4) = s1;
          *(float4*)(red2 + (wave * 16 + lr) * 32 + jj * 16 + lq * 4) = s2;
        }
      }
      if (in_part) {
        __syncthreads();
        if (tid < 32) {
          double t1 = 0.0, t2 = 0.0;
#pragma unroll 8
          for (int g = 0; g < 64; ++g) t1 += (double)red1[g * 32 + tid], t2 += (double)red2[g * 32 + tid];
          double* o = in_part + ((long)tile * 64 + half * 32 + tid) * 2;
          o[0] = t1, o[1] = t2;
        }
      }
    }
  }
}
}  // namespace

int conv_stem_tiles(int H, int W) { return cdiv((H + 6 - 7) / 2 + 1, ST) * cdiv((W + 6 - 7) / 2 + 1, ST); }

// x: f32 NHWC4 frames [nimg][H][W][4]; w: f32 [64][7][7][4] (the fourth input channel's weights are zero); y: f32 [nimg][OH][OW][64];
// in_part: null, or [nimg][conv_stem_tiles(H, W)][64][2] doubles (sum, sum of squares per tile: the layout instnorm_finalize reads)
int conv_stem7x7_x3(const float* x, const float* w, const float* bias, float* y, int nimg, int H, int W, double* in_part, hipStream_t s) {
  if (!x || !w || !y || nimg <= 0 || H < 7 || W < 7) return SAMPT_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias) & 15) return SAMPT_ERR_ARG;
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  const int ntx = cdiv(OW, ST), nty = cdiv(OH, ST);
  const int ldsb = WIMG + 2 * ((PLANE + 15) & ~15);
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute((const void*)k_stem7x7s2_x3, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) != hipSuccess) return SAMPT_ERR_HIP;
    raised = true;
  }
  const long ntiles = (long)nimg * ntx * nty;
  const int grid = (int)(ntiles < 512 ? ntiles : 512);
  hipLaunchKernelGGL(k_stem7x7s2_x3, dim3(grid), dim3(256), ldsb, s, (const float4*)x, w, bias, y, in_part, nimg, H, W, OH, OW, ntx, nty);
  SAMPT_CHECK_LAUNCH("conv_stem7x7_x3");
  return SAMPT_OK;
}

}  // namespace sampt
