// PIPS engine: fnet (BasicEncoder, pips.py:191-287) and the iterative point-update window (pips.py:439-620).
#include <stdlib.h>

#include "engine.h"

namespace sampt {

static int load_conv(const WeightMap& w, const std::string& name, int cin, int cout, int k, int stride, int pad,
                     ConvW& c) {
  c.w = w.f(name + ".weight");
  c.b = w.f(name + ".bias");
  c.w_hl = (cin % 32 == 0 && w.has(name + ".weight_hl")) ? w.h(name + ".weight_hl") : nullptr;
  c.cin = cin, c.cout = cout, c.k = k, c.stride = stride, c.pad = pad;
  return (c.w && c.b) ? SAMPT_OK : SAMPT_ERR_ARG;
}

int PipsEngine::init_fnet(const WeightMap& w) {
  int rc = SAMPT_OK;
  // conv weights arrive repacked [Cout][KH*KW*Cin] (ci fastest); the stem's Cin is zero-padded 3 -> 4
  rc |= load_conv(w, "fnet.conv1", 4, 64, 7, 2, 3, stem);
  const int dims[4] = {64, 96, 128, 128}, strides[4] = {1, 2, 2, 2};
  int in_planes = 64;
  for (int li = 0; li < 4; ++li) {
    for (int bi = 0; bi < 2; ++bi) {
      int cin = bi == 0 ? in_planes : dims[li];
      int st = bi == 0 ? strides[li] : 1;
      std::string p = "fnet.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      rc |= load_conv(w, p + ".conv1", cin, dims[li], 3, st, 1, blk[li][bi][0]);
      rc |= load_conv(w, p + ".conv2", dims[li], dims[li], 3, 1, 1, blk[li][bi][1]);
      has_down[li][bi] = (bi == 0 && st != 1);
      if (has_down[li][bi]) rc |= load_conv(w, p + ".downsample.0", cin, dims[li], 1, st, 0, blk[li][bi][2]);
    }
    in_planes = dims[li];
  }
  rc |= load_conv(w, "fnet.conv2", 416, 256, 3, 1, 1, conv2);
  rc |= load_conv(w, "fnet.conv3", 256, 128, 1, 1, 0, conv3);
  if (rc != SAMPT_OK || !w.missing.empty()) {
    error = "PipsEngine: missing fnet weights: " + w.missing;
    return SAMPT_ERR_ARG;
  }
  return SAMPT_OK;
}

int PipsEngine::init(const WeightMap& w) {
  int rc = init_fnet(w);
  if (rc != SAMPT_OK) return rc;
  const std::string d = "delta_block.to_delta.";
  in_w = w.f(d + "0.weight"), in_b = w.f(d + "0.bias");
  for (int i = 0; i < 12; ++i) {
    std::string p = d + std::to_string(i + 1);
    MixBlk& m = mix[i];
    m.ln1w = w.f(p + ".0.norm.weight"), m.ln1b = w.f(p + ".0.norm.bias");
    m.tw1 = w.f(p + ".0.fn.0.weight"), m.tb1 = w.f(p + ".0.fn.0.bias");
    m.tw2 = w.f(p + ".0.fn.3.weight"), m.tb2 = w.f(p + ".0.fn.3.bias");
    m.ln2w = w.f(p + ".1.norm.weight"), m.ln2b = w.f(p + ".1.norm.bias");
    m.cw1 = w.f(p + ".1.fn.0.weight"), m.cb1 = w.f(p + ".1.fn.0.bias");
    m.cw2 = w.f(p + ".1.fn.3.weight"), m.cb2 = w.f(p + ".1.fn.3.bias");
    m.x3s[0] = w.has(p + ".__x3s16") ? w.h(p + ".__x3s16") : nullptr;
    m.x3s[1] = w.has(p + ".__x3s32") ? w.h(p + ".__x3s32") : nullptr;
  }
  oln_w = w.f(d + "13.weight"), oln_b = w.f(d + "13.bias");
  head_w = w.f(d + "15.weight"), head_b = w.f(d + "15.bias");
  gn_w = w.f("norm.weight"), gn_b = w.f("norm.bias");
  up_wT = w.f("ffeat_updater.0.weight_t"), up_b = w.f("ffeat_updater.0.bias");
  vis_w = w.f("vis_predictor.0.weight"), vis_b = w.f("vis_predictor.0.bias");
  times = w.f("__times");
  if (rc != SAMPT_OK || !w.missing.empty()) {
    error = "PipsEngine: missing weights: " + w.missing;
    return SAMPT_ERR_ARG;
  }
  return SAMPT_OK;
}

// conv (implicit GEMM, bias fused) -> raw output; returns output dims
struct Planes {          // an activation map pre-split into fp16 planes (written by run_inorm), or {null, null}
  half_t *hi = nullptr, *lo = nullptr;
};

struct NormCtx {
  double* partials;
  float* mean_rstd;
  int chunks = 0;          // > 0: the convolution that produced the map already wrote its InstanceNorm partial sums (this many per image)
};

// nc non-null: an InstanceNorm follows — a convolution that can (the halo-tiled 3 x 3 kernel) sums its share of the statistics
static int run_conv(const ConvW& c, const float* x, int n, int H, int W, float* y, int& OH, int& OW, bool dry,
                    hipStream_t s, Planes xp = Planes(), NormCtx* nc = nullptr) {
  OH = (H + 2 * c.pad - c.k) / c.stride + 1;
  OW = (W + 2 * c.pad - c.k) / c.stride + 1;
  if (dry) return SAMPT_OK;
  GemmP p;
  p.A = x, p.W = c.w, p.bias = c.b, p.C = y;
  p.M = n * OH * OW, p.N = c.cout, p.K = c.k * c.k * c.cin;
  p.ldw = p.K, p.ldc = c.cout;
  p.conv = 1, p.cH = H, p.cW = W, p.cC = c.cin, p.KH = c.k, p.KW = c.k, p.cstride = c.stride, p.cpad = c.pad;
  p.OH = OH, p.OW = OW;
  if (c.w_hl) {  // split-fp16 weights packed by the host: fp32-grade result on the fp16 matrix pipe
    p.W = c.w_hl, p.W_lo = c.w_hl + (size_t)c.cout * p.K;
    p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
    if (xp.hi) p.A = xp.hi, p.A_lo = xp.lo;     // activations already split by the producing InstanceNorm
    if (nc && g_conv_halo && (g_conv_in_stats & 1) && conv3x3_halo_eligible(p)) p.in_part = nc->partials, nc->chunks = conv3x3_halo_tiles(p);
    return conv_f16x3(p, s);
  }
  return gemm_f32(p, s);
}

// InstanceNorm (+ReLU) (+skip add + ReLU), in place on y.  planes_only: the normalised map is only ever read by a split-fp16
// convolution (through out.hi / out.lo), so its f32 copy is not written (y keeps the raw convolution output)
static int run_inorm(NormCtx& nc, float* y, int n, long hw, int C, int relu1, const float* skip, bool dry,
                     hipStream_t s, Planes out = Planes(), bool planes_only = false) {
  if (dry) return SAMPT_OK;
  if (nc.chunks > 0) SAMPT_TRY(instnorm_finalize(nc.partials, n, nc.chunks, hw, C, 1e-5f, nc.mean_rstd, s));
  else SAMPT_TRY(instnorm_stats(y, n, hw, C, 1e-5f, nc.partials, nc.mean_rstd, s));
  nc.chunks = 0;
  return instnorm_apply(y, nc.mean_rstd, skip, planes_only && out.hi ? nullptr : y, n, hw, C, relu1, s, out.hi, out.lo);
}

int PipsEngine::fnet(const uint8_t* frames, int nf, int H, int W, float* const out[4], Arena& ws, hipStream_t s) {
  const bool dry = ws.dry();
  const int H2 = (H + 6 - 7) / 2 + 1, W2 = (W + 6 - 7) / 2 + 1;
  NormCtx nc;
  // (chunk partials of instnorm_stats: 512 pixels x <= 256 channels; tile partials of the halo convolution: 16 x 16 pixels x the
  //  layer's channels — 64 at the largest map, 256 at H/4 x W/4)
  size_t tile_part = 0;    // doubles: the largest [nf][16 x 16 tiles][channels][2] any convolution of the encoder writes
  {
    const int dimsl[4] = {64, 96, 128, 128}, stl[4] = {1, 2, 2, 2};
    int hh = H2, ww = W2;
    for (int li = 0; li < 4; ++li) {
      hh = (hh + 2 - 3) / stl[li] + 1, ww = (ww + 2 - 3) / stl[li] + 1;
      tile_part = std::max(tile_part, (size_t)cdiv(hh, 16) * cdiv(ww, 16) * dimsl[li]);
    }
    tile_part = std::max(tile_part, (size_t)cdiv(H2, 16) * cdiv(W2, 16) * 64);
    tile_part = std::max(tile_part, (size_t)cdiv(H / stride, 16) * cdiv(W / stride, 16) * 256) * 2 * nf;
  }
  nc.partials = (double*)ws.get(std::max(instnorm_partial_doubles(nf, (long)H2 * W2, 256), tile_part) * sizeof(double));
  nc.mean_rstd = ws.f32((size_t)nf * 256 * 2);
  NormCtx nc2 = nc;        // a block's conv2 sums its statistics before the downsample branch's InstanceNorm uses nc.partials
  nc2.partials = (double*)ws.get(tile_part * sizeof(double));
  float* x0 = ws.f32((size_t)nf * H * W * 4);
  if (!dry) SAMPT_TRY(rgb_u8chw_to_nhwc4(frames, frames_f32, x0, nf, H, W, s));
  int h, w;
  // Every InstanceNorm output that feeds a split-fp16 convolution is also written as two fp16 planes (same bytes as the
  // f32 map): the convolution then stages ready-made halves instead of splitting each element once per filter tap.
  auto planes = [&](size_t elems, const ConvW& consumer) {
    Planes pl;
    if (consumer.w_hl) pl.hi = ws.f16(elems), pl.lo = ws.f16(elems);
    return pl;
  };
  float* cur = ws.f32((size_t)nf * H2 * W2 * 64);
  Planes cur_p = planes((size_t)nf * H2 * W2 * 64, blk[0][0][0]);
  if (blk[0][0][0].w_hl && g_conv_halo && stem.k == 7 && stem.stride == 2 && stem.pad == 3 && stem.cin == 4 && stem.cout == 64) {
    // split-fp16 mode: the stem as 3-term fp16 products too, one K slab per kernel row (conv_stem_x3.hip), InstanceNorm sums fused
    h = (H + 6 - 7) / 2 + 1, w = (W + 6 - 7) / 2 + 1;
    if (!dry) {
      SAMPT_TRY(conv_stem7x7_x3(x0, stem.w, stem.b, cur, nf, H, W, (g_conv_in_stats & 2) ? nc.partials : nullptr, s));
      if (g_conv_in_stats & 2) nc.chunks = conv_stem_tiles(H, W);
    }
  } else {
    SAMPT_TRY(run_conv(stem, x0, nf, H, W, cur, h, w, dry, s));
  }
  SAMPT_TRY(run_inorm(nc, cur, nf, (long)h * w, 64, 1, nullptr, dry, s, cur_p));
  const int dims[4] = {64, 96, 128, 128};
  float* scale_out[4];
  int sh[4], sw[4];
  for (int li = 0; li < 4; ++li) {
    for (int bi = 0; bi < 2; ++bi) {
      const ConvW& c1 = blk[li][bi][0];
      const ConvW& c2 = blk[li][bi][1];
      int oh, ow, oh2, ow2;
      int ohh = (h + 2 - 3) / c1.stride + 1, oww = (w + 2 - 3) / c1.stride + 1;
      const size_t oel = (size_t)nf * ohh * oww * dims[li];
      float* y1 = ws.f32(oel);
      float* y2 = ws.f32(oel);
      Planes y1_p = planes(oel, c2);
      // the block's output feeds the next block's conv1 (and its 1x1 downsample); the last block's only the resize
      const bool last = li == 3 && bi == 1;
      Planes y2_p = last ? Planes() : planes(oel, bi == 0 ? blk[li][1][0] : blk[li + 1][0][0]);
      SAMPT_TRY(run_conv(c1, cur, nf, h, w, y1, oh, ow, dry, s, cur_p, &nc));
      SAMPT_TRY(run_inorm(nc, y1, nf, (long)oh * ow, dims[li], 1, nullptr, dry, s, y1_p, true));   // y1 feeds conv2 only
      SAMPT_TRY(run_conv(c2, y1, nf, oh, ow, y2, oh2, ow2, dry, s, y1_p, &nc2));
      const float* skip = cur;
      if (has_down[li][bi]) {
        float* dn = y1;  // y1 is dead after conv2 has consumed it (stream order)
        int dh, dw;
        SAMPT_TRY(run_conv(blk[li][bi][2], cur, nf, h, w, dn, dh, dw, dry, s, cur_p));
        SAMPT_TRY(run_inorm(nc, dn, nf, (long)dh * dw, dims[li], 0, nullptr, dry, s));
        skip = dn;
      }
      SAMPT_TRY(run_inorm(nc2, y2, nf, (long)oh2 * ow2, dims[li], 1, skip, dry, s, y2_p));
      cur = y2, cur_p = y2_p, h = oh2, w = ow2;
    }
    scale_out[li] = cur, sh[li] = h, sw[li] = w;
  }
  const int H4 = H / stride, W4 = W / stride;
  // the 4 scales resized to H/4 x W/4 and concatenated (pips.py:266-281): written straight as split fp16 planes when the
  // 416 -> 256 convolution takes them (same bytes as the f32 map, which is then never materialised)
  Planes cat_p = planes((size_t)nf * H4 * W4 * 416, conv2);
  float* cat = cat_p.hi ? nullptr : ws.f32((size_t)nf * H4 * W4 * 416);
  const int coff[4] = {0, 64, 160, 288};
  if (!dry)
    for (int li = 0; li < 4; ++li)
      SAMPT_TRY(resize_bilinear_nhwc(scale_out[li], nf, sh[li], sw[li], dims[li], cat, H4, W4, 416, coff[li], 1, s, cat_p.hi,
                                     cat_p.lo));
  float* y = ws.f32((size_t)nf * H4 * W4 * 256);
  int oh, ow;
  Planes y_p = planes((size_t)nf * H4 * W4 * 256, conv3);
  SAMPT_TRY(run_conv(conv2, cat, nf, H4, W4, y, oh, ow, dry, s, cat_p, &nc));
  SAMPT_TRY(run_inorm(nc, y, nf, (long)oh * ow, 256, 1, nullptr, dry, s, y_p, true));             // feeds the 1 x 1 conv3 only
  SAMPT_TRY(run_conv(conv3, y, nf, oh, ow, out[0], oh, ow, dry, s, y_p));
  if (!dry) {
    int ph = H4, pw = W4;
    for (int l = 1; l < 4; ++l) {
      SAMPT_TRY(avgpool2x2_nhwc(out[l - 1], nf, ph, pw, 128, out[l], s));
      ph /= 2, pw /= 2;
    }
  }
  return ws.ok() ? SAMPT_OK : SAMPT_ERR_WORKSPACE;
}

static int lin(const float* A, int M, int K, int lda, const float* W, const float* b, float* C, int N, int act,
               const float* res, hipStream_t s, float* skws = nullptr, size_t skn = 0) {
  GemmP p;
  p.A = A, p.W = W, p.bias = b, p.C = C, p.res = res;
  p.splitk_ws = skws, p.splitk_ws_floats = skn;
  p.M = M, p.N = N, p.K = K, p.lda = lda, p.ldw = K, p.ldc = N, p.ldr = N, p.act = act;
  return gemm_f32(p, s);
}

#define PIPS_LAUNCH(expr) do { ++nl; SAMPT_TRY(expr); } while (0)     // every kernel launch of a window is counted where it is made

int PipsEngine::update(const PyramidLevels& pyr, const int* frame_idx, int n, const float* xys, const float* feat_init,
                       int iters, float* traj_out, float* vis_out, Arena& ws, hipStream_t s) {
  const bool dry = ws.dry();
  const int LDX = 520, D = 512, R = n * S;
  float* coords = ws.f32((size_t)S * n * 2);
  float* coords0 = ws.f32((size_t)n * 2);
  float* ffeats = ws.f32((size_t)R * 128);
  float* x = ws.f32((size_t)R * LDX);
  float* hbuf = ws.f32((size_t)R * D);
  float* hbuf2 = ws.f32((size_t)R * D);
  float* lnb = ws.f32((size_t)R * D);
  float* hid = ws.f32((size_t)R * 4 * D);
  float* mean = ws.f32((size_t)n * D);
  float* delta = ws.f32((size_t)n * S * 130);
  // fused mixer (pips_mixer.hip): one slab [R][512] per hidden slice; always carved, so the size does not depend on the knob
  float* part = ws.f32((size_t)32 * R * D);
  half_t* xop = ws.f16(pips_mix_xop_halves(n));
  if (!ws.ok()) return SAMPT_ERR_WORKSPACE;
  if (dry) return SAMPT_OK;
  int nl = 0;                                  // kernel launches of this window (counted at the launch sites)
  const bool fused = g_pips_mixer_fused != 0;
  const int NS = pips_mix_slices(n);
  // split-fp16 channel MLP: when asked for and the packer delivered the operand streams (weights outside the splittable range: f32)
  const int x3i = g_pips_mixer_wgs >= 32 ? 1 : 0, NSx = x3i ? 32 : 16;
  bool x3 = fused && g_pips_mixer_x3 != 0;
  for (int i = 0; i < 12 && x3; ++i) x3 = mix[i].x3s[x3i] != nullptr;
  PIPS_LAUNCH(pips_init_state(xys, feat_init, (float)stride, S, n, coords, coords0, ffeats, s));
  for (int it = 0; it < iters; ++it) {
    if (x3) {
      // the same 30 launches with the channel MLP as split-fp16 products (pips_mixer_x3.hip): [slab sum + residual -> token
      // mixing -> LayerNorm2 -> operand images] and [fc1 -> GELU -> fc2 slabs from the packed weight stream]
      PIPS_LAUNCH(pips_corr_sample(pyr, frame_idx, S, n, 128, ffeats, coords, x, LDX, 128, s, times));
      PIPS_LAUNCH(lin(x, R, LDX, LDX, in_w, in_b, hbuf, D, ACT_NONE, nullptr, s));
      float* xpp[2] = {hbuf2, lnb};
      const float* prev = hbuf;
      for (int i = 0; i < 12; ++i) {
        const MixBlk& m = mix[i];
        PIPS_LAUNCH(pips_mix_pre(i ? part : nullptr, i ? NSx : 0, i ? mix[i - 1].cb2 : nullptr, prev, n, m.ln1w, m.ln1b, m.tw1,
                                 m.tb1, m.tw2, m.tb2, m.ln2w, m.ln2b, xpp[i & 1], xop, s));
        PIPS_LAUNCH(pips_mix_mlp_x3(xop, m.x3s[x3i], m.cb1, part, n, NSx, s));
        prev = xpp[i & 1];
      }
      PIPS_LAUNCH(pips_mix_reduce(part, NSx, mix[11].cb2, prev, n, 1, oln_w, oln_b, nullptr, nullptr, nullptr, nullptr, mean, s));
    } else if (fused) {
      // 30 launches per iteration: input (1), in-projection (1), per block [sum of the previous block's slabs + residual ->
      // token mixing] and [LayerNorm -> fc1 -> GELU -> fc2 slabs] (2 x 12), last sum + LayerNorm + token mean (1), head, update
      PIPS_LAUNCH(pips_corr_sample(pyr, frame_idx, S, n, 128, ffeats, coords, x, LDX, 128, s, times));
      PIPS_LAUNCH(lin(x, R, LDX, LDX, in_w, in_b, hbuf, D, ACT_NONE, nullptr, s));
      float* xpp[2] = {hbuf2, lnb};
      const float* prev = hbuf;
      for (int i = 0; i < 12; ++i) {
        const MixBlk& m = mix[i];
        PIPS_LAUNCH(pips_mix_reduce(i ? part : nullptr, i ? NS : 0, i ? mix[i - 1].cb2 : nullptr, prev, n, 0, m.ln1w, m.ln1b,
                                  m.tw1, m.tb1, m.tw2, m.tb2, xpp[i & 1], s));
        PIPS_LAUNCH(pips_mix_mlp(xpp[i & 1], m.ln2w, m.ln2b, m.cw1, m.cb1, m.cw2, part, n, NS, s));
        prev = xpp[i & 1];
      }
      PIPS_LAUNCH(pips_mix_reduce(part, NS, mix[11].cb2, prev, n, 1, oln_w, oln_b, nullptr, nullptr, nullptr, nullptr, mean, s));
    } else {
      PIPS_LAUNCH(pips_corr_sample(pyr, frame_idx, S, n, 128, ffeats, coords, x, LDX, 128, s));
      PIPS_LAUNCH(pips_build_input(ffeats, coords, times, S, n, x, LDX, s));
      PIPS_LAUNCH(lin(x, R, LDX, LDX, in_w, in_b, hbuf, D, ACT_NONE, nullptr, s));
      // 12 mixer blocks, 4 launches each: token mixing, LayerNorm, fc1 + GELU, fc2 + residual (thin GEMMs: K split inside
      // the workgroup, no split-K grid + reduction pass).
      for (int i = 0; i < 12; ++i) {
        const MixBlk& m = mix[i];
        PIPS_LAUNCH(pips_token_mix(hbuf, hbuf2, m.ln1w, m.ln1b, m.tw1, m.tb1, m.tw2, m.tb2, n, S, D, s));
        PIPS_LAUNCH(layernorm_rows(hbuf2, m.ln2w, m.ln2b, lnb, R, D, 1e-5f, nullptr, 0, ACT_NONE, s));
        PIPS_LAUNCH(lin(lnb, R, D, D, m.cw1, m.cb1, hid, 4 * D, ACT_GELU, nullptr, s));
        PIPS_LAUNCH(lin(hid, R, 4 * D, 4 * D, m.cw2, m.cb2, hbuf, D, ACT_NONE, hbuf2, s));
      }
      PIPS_LAUNCH(pips_ln_mean(hbuf, oln_w, oln_b, mean, n, S, D, s));
    }
    PIPS_LAUNCH(lin(mean, n, D, D, head_w, head_b, delta, S * 130, ACT_NONE, nullptr, s));
    PIPS_LAUNCH(pips_update(delta, gn_w, gn_b, up_wT, up_b, ffeats, coords, coords0, S, n, s));
  }
  PIPS_LAUNCH(pips_finalize(ffeats, vis_w, vis_b, coords, (float)stride, S, n, traj_out, vis_out, s));
  window_launches = nl;
  return SAMPT_OK;
}
#undef PIPS_LAUNCH

int PipsEngine::track(const PyramidLevels& pyr, int T, int n, const float* q, const unsigned char* flip, const float* q_host,
                      const unsigned char* flip_host, float thr0, int iters, void* const* chunk_ev, const int* chunk_lo,
                      const int* chunk_hi, int nchunks, int* flag, hipEvent_t flag_ev[2], float* traj, float* vis, Arena& ws,
                      hipStream_t s, int* rounds) {
  const bool dry = ws.dry();
  int* cur = (int*)ws.get((size_t)n * sizeof(int));
  int* fidx = (int*)ws.get((size_t)n * S * sizeof(int));
  int* f0 = (int*)ws.get((size_t)n * sizeof(int));
  int* n_active = (int*)ws.get(256);
  float* xys = ws.f32((size_t)n * 2);
  float* xy_feat = ws.f32((size_t)n * 2);
  float* feat_init = ws.f32((size_t)n * 128);
  float* tr = ws.f32((size_t)S * n * 2);
  float* vi = ws.f32((size_t)S * n);
  if (dry) {   // the window's own scratch, measured
    SAMPT_TRY(update(pyr, nullptr, n, nullptr, nullptr, iters, nullptr, nullptr, ws, s));
    return ws.ok() ? SAMPT_OK : SAMPT_ERR_WORKSPACE;
  }
  if (!ws.ok()) return SAMPT_ERR_WORKSPACE;
  const size_t ws_mark = ws.off;
  // host view: the frames a round can reach.  A chain anchored at frame a reads [a, a + S - 1] and its next anchor is at
  // most a + S - 1, so after r rounds every anchor is <= start + r (S - 1).
  int smin[2] = {T, T}, smax[2] = {-1, -1};
  for (int i = 0; i < n; ++i) {
    const int t0 = (int)q_host[i * 3], d = flip_host[i] ? 1 : 0;
    if (t0 >= T - 1) continue;                 // never active (tracker.py:67)
    smin[d] = t0 < smin[d] ? t0 : smin[d];
    smax[d] = t0 > smax[d] ? t0 : smax[d];
  }
  std::vector<char> waited(nchunks > 0 ? nchunks : 0, 0);
  auto wait_chunks = [&](int r) -> int {
    for (int d = 0; d < 2; ++d) {
      if (smax[d] < 0) continue;
      int lo = smin[d], hi = smax[d] + (r + 1) * (S - 1);
      if (hi > T - 1) hi = T - 1;
      if (d) { const int a = T - 1 - hi, b = T - 1 - lo; lo = a, hi = b; }
      for (int c = 0; c < nchunks; ++c)
        if (!waited[c] && chunk_lo[c] <= hi && chunk_hi[c] > lo) {
          if (hipStreamWaitEvent(s, (hipEvent_t)chunk_ev[c], 0) != hipSuccess) return SAMPT_ERR_HIP;
          waited[c] = 1;
        }
    }
    return SAMPT_OK;
  };
  *rounds = 0;
  SAMPT_TRY(pips_chain_init(q, n, T, cur, traj, vis, s));
  if (smax[0] < 0 && smax[1] < 0) return SAMPT_OK;                       // every query sits on its direction's last frame
  const int max_rounds = T + 1;
  for (int r = 0; r < max_rounds; ++r) {
    SAMPT_TRY(wait_chunks(r));
    SAMPT_TRY(pips_round_begin(cur, flip, traj, T, n, S, fidx, xys, r == 0 ? xy_feat : nullptr, r == 0 ? f0 : nullptr,
                               (float)stride, s));
    if (r == 0)   // tracker.py:81-90 == App. B-6: the only used output of the init pass is the feature at the query frame
      SAMPT_TRY(pips_sample_feat(pyr.base[0], pyr.H[0], pyr.W[0], 128, f0, xy_feat, n, feat_init, s));
    ws.off = ws_mark;
    SAMPT_TRY(update(pyr, fidx, n, xys, feat_init, iters, tr, vi, ws, s));
    SAMPT_TRY(pips_round_end(cur, tr, vi, T, n, S, thr0, traj, vis, n_active, s));
    if (hipMemcpyAsync(&flag[r & 1], n_active, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return SAMPT_ERR_HIP;
    if (hipEventRecord(flag_ev[r & 1], s) != hipSuccess) return SAMPT_ERR_HIP;
    ++*rounds;
    if (r >= 1) {   // one round of look-ahead: the host enqueues round r while the device runs it / finishes round r - 1
      if (hipEventSynchronize(flag_ev[(r - 1) & 1]) != hipSuccess) return SAMPT_ERR_HIP;
      if (flag[(r - 1) & 1] == 0) break;       // round r found no active chain and wrote nothing
    }
  }
  if (hipStreamSynchronize(s) != hipSuccess) return SAMPT_ERR_HIP;
  if (flag[0] != 0 && flag[1] != 0) {
    error = "PipsEngine::track: chains still active after T + 1 rounds";
    return SAMPT_ERR_ARG;
  }
  return SAMPT_OK;
}

}  // namespace sampt
