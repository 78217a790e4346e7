// SAM image encoder (ImageEncoderViT) as a fixed launch sequence.
//   fast mode (c.f16 = 1): fp16 MFMA GEMMs with fused bias / GELU / residual epilogues, fused flash attention with
//                          on-the-fly decomposed rel-pos bias, fp32 residual stream / LayerNorm / softmax.
//   exact mode (c.f16 = 0): the same graph in fp32 (f32 MFMA GEMMs, materialised scores) — parity reference.
//   f16x3 mode (c.f16 = 2): the fast mode's graph with every product rebuilt from split-fp16 pieces (3 fp16 MFMAs per
//                          product, fp32 accumulate): fp32-grade results at a third of the fp16 MFMA rate instead of the f32
//                          MFMA's 1/16.  Activations travel between kernels as "x3 rows" (common.h GemmP::x3): LayerNorm,
//                          the qkv / fc1 GEMM epilogues and the attention kernel emit them, the next GEMM consumes them.
// Activations are token-major [rows][channels] throughout (NHWC): the neck's output is directly the decoder's
// image-token matrix.
#include "engine.h"

namespace sampt {

int VitEngine::init(const WeightMap& w, const VitConfig& cfg, int wrb) {
  c = cfg;
  win_rows_batches = wrb;
  const std::string sfx = c.f16 == 2 ? ".x3" : (c.f16 ? ".f16" : "");
  const std::string e = "image_encoder.";
  // the encoder's two ends are fp32-grade in both modes (see encode): exact f32 weights, or their split-fp16 planes
  patch_hl = c.f16 ? w.h(e + "patch_embed.proj.weight_hl") : nullptr;
  neck0_hl = c.f16 ? w.h(e + "neck.0.weight_hl") : nullptr;
  neck2_hl = c.f16 ? w.h(e + "neck.2.weight_khwc_hl") : nullptr;
  patch_w = c.f16 ? nullptr : w.get(e + "patch_embed.proj.weight");
  patch_b = w.f(e + "patch_embed.proj.bias");
  pos = w.f(e + "pos_embed");
  blk.resize(c.depth);
  for (int i = 0; i < c.depth; ++i) {
    std::string p = e + "blocks." + std::to_string(i);
    Blk& b = blk[i];
    b.ln1w = w.f(p + ".norm1.weight"), b.ln1b = w.f(p + ".norm1.bias");
    b.ln2w = w.f(p + ".norm2.weight"), b.ln2b = w.f(p + ".norm2.bias");
    b.qkv_w = w.get(p + ".attn.qkv.weight" + sfx), b.qkv_b = w.f(p + ".attn.qkv.bias");
    if (c.f16) b.qkv_b16 = w.h(p + ".attn.qkv.bias" + sfx);
    b.proj_w = w.get(p + ".attn.proj.weight" + sfx), b.proj_b = w.f(p + ".attn.proj.bias");
    b.rel_h = w.f(p + ".attn.rel_pos_h"), b.rel_w = w.f(p + ".attn.rel_pos_w");
    b.rel_ops = w.has(p + ".attn.rel_pos_ops") ? w.h(p + ".attn.rel_pos_ops") : nullptr;
    b.w1 = w.get(p + ".mlp.lin1.weight" + sfx), b.b1 = w.f(p + ".mlp.lin1.bias");
    b.w2 = w.get(p + ".mlp.lin2.weight" + sfx), b.b2 = w.f(p + ".mlp.lin2.bias");
  }
  neck0_w = c.f16 ? nullptr : w.get(e + "neck.0.weight");
  neck1w = w.f(e + "neck.1.weight"), neck1b = w.f(e + "neck.1.bias");
  neck2_w = c.f16 ? nullptr : w.get(e + "neck.2.weight_khwc");  // repacked [Cout][ky][kx][Cin]
  neck3w = w.f(e + "neck.3.weight"), neck3b = w.f(e + "neck.3.bias");
  win_rows = w.i("__win_rows");
  win_inv = w.i("__win_inv"), win_pad = w.i("__win_pad");
  for (int k = 0; k < 8; ++k) {   // window maps of the compact live grids (k+1 window rows x full width), optional
    const std::string sk = std::to_string(k + 1);
    win_inv_live[k] = w.has("__win_inv_live" + sk) ? w.i("__win_inv_live" + sk) : nullptr;
    win_pad_live[k] = w.has("__win_pad_live" + sk) ? w.i("__win_pad_live" + sk) : nullptr;
  }
  if (!w.missing.empty()) {
    error = "VitEngine: missing weights: " + w.missing;
    return SAMPT_ERR_ARG;
  }
  return SAMPT_OK;
}

int VitEngine::profile_end(double* flop, double* ms, int* launches) {
  double f = 0.0, t = 0.0;
  for (auto& e : prof) {
    float dt = 0.f;
    if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&dt, e.a, e.b) != hipSuccess) return SAMPT_ERR_HIP;
    f += e.flop, t += dt;
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  *flop = f, *ms = t, *launches = (int)prof.size();
  prof.clear();
  profiling = false;
  return SAMPT_OK;
}

namespace {
struct G {
  int mode;            // VitConfig::f16: 0 exact f32, 1 fp16, 2 split-fp16 (x3 rows)
  hipStream_t s;
  const VitEngine* eng = nullptr;
  // C = act(A.W^T + bias) (+ residual at the (row-mapped) destination row).  out: 0 = f32, 1 = fp16, 2 = x3 rows.
  // M, N, K are the logical sizes; in mode 2 A and W are x3 rows (2K halves per row).
  int run(const void* A, int M, int K, const void* W, const float* bias, void* C, int N, int act, int out,
          const float* res, int ldr, const int* rowmap, int res_mod, const int* a_rowmap = nullptr,
          bool exact = false, const half_t* hl = nullptr, int kind = -1) const {
    GemmP p;
    p.A = A, p.W = W, p.bias = bias, p.C = C, p.res = res, p.rowmap = rowmap, p.a_rowmap = a_rowmap;
    p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldc = N, p.ldr = ldr, p.act = act, p.res_mod = res_mod;
    p.out_f16 = out;
    p.p8_wgs = eng ? (kind >= 0 && eng->gemm_wgs_kind[kind] > 0 ? eng->gemm_wgs_kind[kind] : eng->gemm_wgs) : 0;
    if (exact && hl) {   // fp32-grade on the fp16 pipe: the GEMM as a 1x1 convolution over an [1][M][1][K] image
      p.W = hl, p.W_lo = hl + (size_t)N * K;
      p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
      p.conv = 1, p.cH = M, p.cW = 1, p.cC = K, p.KH = 1, p.KW = 1, p.cstride = 1, p.cpad = 0, p.OH = M, p.OW = 1;
      return conv_f16x3(p, s);
    }
    if (mode == 0 || exact) return gemm_f32(p, s);
    if (mode == 2) {
      p.x3 = 1, p.K = 2 * K, p.lda = 2 * K, p.ldw = 2 * K;
      p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
      if (out == 2) p.ldc = 2 * N;
    }
    if (eng && eng->calib && kind >= 0 && mode == 1 && K <= eng->calib_ld)
      SAMPT_TRY(colmean_rows_f16((const half_t*)A, M, K, K, a_rowmap, eng->calib + ((size_t)eng->cur_blk * 4 + kind) * eng->calib_ld, s));
    if (!eng || !eng->profiling) return gemm_f16(p, s);
    VitEngine::GemmEv ev;
    ev.flop = 2.0 * M * N * K;
    if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return SAMPT_ERR_HIP;
    if (hipEventRecord(ev.a, s) != hipSuccess) return SAMPT_ERR_HIP;
    int rc = gemm_f16(p, s);
    if (hipEventRecord(ev.b, s) != hipSuccess) return SAMPT_ERR_HIP;
    eng->prof.push_back(ev);
    return rc;
  }
};
}  // namespace

int VitEngine::live_rows(int H, int W) const {
  // Token rows that can depend on the frame before the first global-attention block: the rows holding pixels, completed to
  // whole windows.  Only for landscape / square-width frames (the live tokens are then a contiguous prefix of every frame).
  int g0 = 0;
  while (g0 < c.depth && !((c.global_mask >> g0) & 1)) ++g0;
  if (g0 == 0 || W != c.img) return c.grid;
  const int h_tok = (H + c.patch - 1) / c.patch;
  const int lh = ((h_tok + c.window - 1) / c.window) * c.window;
  if (lh >= c.grid || lh / c.window > 8 || !win_inv_live[lh / c.window - 1]) return c.grid;
  return lh;
}

int VitEngine::encode(const uint8_t* frames, int chw, int B, int H, int W, float* features, float* interm_out, Arena& ws,
                      hipStream_t s, float* dead_cache, int dead_mode) {
  const bool dry = ws.dry();
  const int g = c.grid, T = g * g, D = c.D, ws_ = c.window, hd = D / c.heads;
  const int gp = ((g + ws_ - 1) / ws_) * ws_, nw1 = gp / ws_, nwin = nw1 * nw1, wt = ws_ * ws_;
  const long Mg = (long)B * T, Mw = (long)B * nwin * wt, Mmax = Mw > Mg ? Mw : Mg;
  const size_t esz = c.f16 == 1 ? 2 : 4;                // fp16 rows; f32 rows and x3 rows (2 halves per element) alike
  const int Kp = 3 * c.patch * c.patch;
  if (B > win_rows_batches) return SAMPT_ERR_ARG;

  // ---- frames whose height is not the padded square's: the token rows below the picture (zero padding after
  //      Sam.preprocess, 28 of 64 rows for 16:9 video) hold the same values in EVERY frame until the first global-attention
  //      block mixes them with picture tokens — their patch embedding is bias + pos_embed and the windowed blocks only mix
  //      tokens of one window.  Blocks before the first global one therefore run on a compact stream of the live rows only
  //      (dead_mode 2) and the dead rows' residual stream is taken from a cache computed once per frame geometry (dead_mode
  //      1: one zero frame through the full path).  Bit-identical to the full computation: GEMM rows, LayerNorm rows and
  //      attention windows are independent of which other rows / windows share the launch.
  int g0 = 0;
  while (g0 < c.depth && !((c.global_mask >> g0) & 1)) ++g0;
  const int lh = live_rows(H, W);
  const bool rect = dead_mode == 2 && lh < g;
  if (dead_mode != 0 && (lh >= g || !dead_cache)) return SAMPT_ERR_ARG;
  if (dead_mode == 1 && B != 1) return SAMPT_ERR_ARG;
  // calibration (sampt_vit_calibrate) records ONE frame's column means through the plain path: anything else would average
  // over frames or over the compacted live rows only, silently
  if (calib && !ws.dry() && (B != 1 || dead_mode != 0)) {
    error = "VitEngine::encode: calibration is set (sampt_vit_calibrate): one frame through the plain entry point only";
    return SAMPT_ERR_ARG;
  }
  const int Tl = lh * g, nwin_l = (lh / ws_) * nw1;
  const long Ml = (long)B * Tl;
  float* xl = rect ? ws.f32((size_t)Ml * D) : nullptr;
  float* x = ws.f32((size_t)Mg * D);
  size_t xn_bytes = (size_t)Mmax * (D > Kp ? D : Kp) * esz;        // LN output / attention output (fp16 in the fast mode)
  if ((size_t)Mg * Kp * 4 > xn_bytes) xn_bytes = (size_t)Mg * Kp * 4;  // ... and the fp32 patch matrix
  void* xn = ws.get(xn_bytes);
  void* qkv = ws.get((size_t)Mmax * 3 * D * esz);
  void* att = ws.get((size_t)Mmax * D * esz);
  void* hid = ws.get((size_t)Mg * c.mlp_ratio * D * esz);
  const int Smax = g > ws_ ? g : ws_;
  const long BHg = (long)B * c.heads, BHw = (long)B * nwin * c.heads;
  size_t rel_elems = (size_t)BHg * g * T;
  if ((size_t)BHw * ws_ * wt > rel_elems) rel_elems = (size_t)BHw * ws_ * wt;
  float* relh = c.f16 ? nullptr : ws.f32(rel_elems);   // materialised bias tables: exact-fp32 mode only
  float* relw = c.f16 ? nullptr : ws.f32(rel_elems);
  float* scores = nullptr;
  if (!c.f16) {
    size_t sg = (size_t)BHg * T * T, sw2 = (size_t)BHw * wt * wt;
    scores = ws.f32(sg > sw2 ? sg : sw2);
  }
  float* neck_a = ws.f32((size_t)Mg * c.out_chans);
  float* neck_b = ws.f32((size_t)Mg * c.out_chans);
  if (!ws.ok()) return SAMPT_ERR_WORKSPACE;
  if (dry) return SAMPT_OK;
  (void)Smax;

  G gm{c.f16, s, this};
  const int act_out = c.f16;                             // GEMM outputs that feed the next GEMM / attention: f32, fp16 or x3 rows
  // ---- patch embedding: preprocess + im2col, GEMM + bias + positional embedding (broadcast over the batch)
  //      (exact fp32 in both modes: first and last layers of the encoder, 0.3 % of its FLOPs)
  SAMPT_TRY(sam_patchify(frames, chw, B, H, W, c.img, c.patch, c.mean, c.stdv, xn, 0, s));
  SAMPT_TRY(gm.run(xn, (int)Mg, Kp, patch_w, patch_b, x, D, ACT_NONE, 0, pos, D, nullptr, T, nullptr, true, patch_hl));

  const float scale = 1.0f / sqrtf((float)hd);
  bool tapped = false;
  float* const x_full = x;
  const long Mg_full = Mg;
  if (rect) {   // gather the live rows of every frame into the compact stream
    if (hipMemcpy2DAsync(xl, (size_t)Tl * D * 4, x_full, (size_t)T * D * 4, (size_t)Tl * D * 4, B, hipMemcpyDeviceToDevice, s) !=
        hipSuccess)
      return SAMPT_ERR_HIP;
  }
  for (int i = 0; i < c.depth; ++i) {
    const Blk& b = blk[i];
    cur_blk = i;
    const bool glob = (c.global_mask >> i) & 1;
    const bool compact = rect && i < g0;
    if (i == g0 && (rect || dead_mode == 1)) {
      if (dead_mode == 1) {   // the zero frame went through the full path: keep its dead rows, done
        if (hipMemcpyAsync(dead_cache, x_full + (size_t)Tl * D, (size_t)(T - Tl) * D * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
          return SAMPT_ERR_HIP;
        return SAMPT_OK;
      }
      // back to the full token grid: live rows from the compact stream, dead rows from the cache
      if (hipMemcpy2DAsync(x_full, (size_t)T * D * 4, xl, (size_t)Tl * D * 4, (size_t)Tl * D * 4, B, hipMemcpyDeviceToDevice, s) !=
          hipSuccess)
        return SAMPT_ERR_HIP;
      for (int f = 0; f < B; ++f)
        if (hipMemcpyAsync(x_full + ((size_t)f * T + Tl) * D, dead_cache, (size_t)(T - Tl) * D * 4, hipMemcpyDeviceToDevice, s) !=
            hipSuccess)
          return SAMPT_ERR_HIP;
    }
    float* const x = compact ? xl : x_full;               // (shadows: everything below works on the active stream)
    const long Mg = compact ? Ml : Mg_full;
    const int S = glob ? g : ws_, N = S * S;
    const int Bw = glob ? B : B * (compact ? nwin_l : nwin);
    const long M = (long)Bw * N;
    const int* inv = glob ? nullptr : (compact ? win_inv_live[lh / ws_ - 1] : win_inv);
    const int* win_pad = compact ? win_pad_live[lh / ws_ - 1] : this->win_pad;
    // norm1; the window partition (zero padding AFTER the norm, App. A-3) is a row scatter of the qkv GEMM: only the
    // real tokens go through the GEMM, the padded rows' qkv is the bias alone
    SAMPT_TRY(layernorm_rows(x, b.ln1w, b.ln1b, xn, Mg, D, 1e-6f, nullptr, c.f16, ACT_NONE, s));
    SAMPT_TRY(gm.run(xn, (int)Mg, D, b.qkv_w, b.qkv_b, qkv, 3 * D, ACT_NONE, act_out, nullptr, 0, inv, 0, nullptr, false, nullptr, 0));
    // the 16-bit attention kernels take the padded tokens' K / V from the bias row themselves (FlashPad); only the exact mode's
    // materialised path needs the padded qkv rows written
    FlashPad fp;
    if (!glob && c.f16) {
      fp.bias_row = b.qkv_b16, fp.nwx = nw1, fp.nwin = compact ? nwin_l : nwin, fp.gh = compact ? lh : g, fp.gw = g;
    } else if (!glob) {
      SAMPT_TRY(fill_rows_bias(qkv, 0, win_pad, (int)(M - Mg), b.qkv_b, 3 * D, s));
    }
    if (c.f16 == 1) {
      fp.rel_ops = b.rel_ops;
      SAMPT_TRY(vit_flash_attention_f16((const half_t*)qkv, b.rel_h, b.rel_w, (half_t*)att, Bw, S, c.heads, hd, s, fp));
    } else if (c.f16 == 2) {
      SAMPT_TRY(vit_flash_attention_x3((const half_t*)qkv, b.rel_h, b.rel_w, (half_t*)att, Bw, S, c.heads, hd, s, fp));
    } else {
      SAMPT_TRY(vit_rel_bias(qkv, 0, b.rel_h, b.rel_w, Bw, S, c.heads, hd, relh, relw, s));
      const float* q = (const float*)qkv;
      GemmP p;  // scores[bw][h] = scale * Q K^T
      p.A = q, p.W = q + D, p.C = scores;
      p.M = N, p.N = N, p.K = hd, p.lda = 3 * D, p.ldw = 3 * D, p.ldc = N;
      p.nb1 = Bw, p.nb2 = c.heads;
      p.sA1 = (long)N * 3 * D, p.sA2 = hd, p.sW1 = (long)N * 3 * D, p.sW2 = hd;
      p.sC1 = (long)c.heads * N * N, p.sC2 = (long)N * N;
      p.alpha = scale;
      SAMPT_TRY(gemm_f32(p, s));
      SAMPT_TRY(softmax_rel_rows(scores, relh, relw, (long)Bw * c.heads, N, S, s));
      GemmP v;  // out[bw][:, h*hd:(h+1)*hd] = P V
      v.A = scores, v.W = q + 2 * D, v.C = att, v.w_kn = 1;
      v.M = N, v.N = hd, v.K = N, v.lda = N, v.ldw = 3 * D, v.ldc = D;
      v.nb1 = Bw, v.nb2 = c.heads;
      v.sA1 = (long)c.heads * N * N, v.sA2 = (long)N * N, v.sW1 = (long)N * 3 * D, v.sW2 = hd;
      v.sC1 = (long)N * D, v.sC2 = hd;
      SAMPT_TRY(gemm_f32(v, s));
    }
    // proj + bias + residual on the real tokens: A rows are gathered from the window-ordered attention output
    SAMPT_TRY(gm.run(att, (int)Mg, D, b.proj_w, b.proj_b, x, D, ACT_NONE, 0, x, D, nullptr, 0, inv, false, nullptr, 1));
    // MLP
    SAMPT_TRY(layernorm_rows(x, b.ln2w, b.ln2b, xn, Mg, D, 1e-6f, nullptr, c.f16, ACT_NONE, s));
    SAMPT_TRY(gm.run(xn, (int)Mg, D, b.w1, b.b1, hid, c.mlp_ratio * D, ACT_GELU, act_out, nullptr, 0, nullptr, 0, nullptr, false,
                     nullptr, 2));
    SAMPT_TRY(gm.run(hid, (int)Mg, c.mlp_ratio * D, b.w2, b.b2, x, D, ACT_NONE, 0, x, D, nullptr, 0, nullptr, false, nullptr, 3));
    if (interm_out && glob && !tapped) {  // HQ-SAM: the first global block's output feeds compress_vit_feat
      if (hipMemcpyAsync(interm_out, x, (size_t)Mg * D * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return SAMPT_ERR_HIP;
      tapped = true;
    }
  }
  if (dead_mode == 1) return SAMPT_ERR_ARG;   // (no global block: live_rows() already refused)
  // ---- neck: conv1x1 (no bias) -> LayerNorm2d -> conv3x3 (no bias) -> LayerNorm2d
  //      fp32 in both modes: the neck's operand roundings would land on the embedding undamped
  SAMPT_TRY(gm.run(x, (int)Mg, D, neck0_w, nullptr, neck_a, c.out_chans, ACT_NONE, 0, nullptr, 0, nullptr, 0, nullptr,
                   true, neck0_hl));
  // (split-fp16 neck: the first LayerNorm2d writes its output as the two fp16 planes the halo-tiled 3 x 3 convolution stages by
  //  LDS-DMA — same bytes as the f32 map in the same buffer, same hi / lo values the tiled kernel split on the fly, ~165 -> 110 us
  //  per 8 frames (profiles/r6_c40_*); conv_halo_x3.hip)
  const bool planes = neck2_hl && g_conv_halo && c.out_chans % 256 == 0 && c.out_chans <= 1536;
  SAMPT_TRY(layernorm_rows(neck_a, neck1w, neck1b, neck_b, Mg, c.out_chans, 1e-6f, nullptr, planes ? 3 : 0, ACT_NONE, s));
  {
    GemmP p;
    p.A = neck_b, p.W = neck2_w, p.C = neck_a;
    p.M = (int)Mg, p.N = c.out_chans, p.K = 9 * c.out_chans, p.ldw = p.K, p.ldc = c.out_chans;
    p.conv = 1, p.cH = g, p.cW = g, p.cC = c.out_chans, p.KH = 3, p.KW = 3, p.cstride = 1, p.cpad = 1, p.OH = g, p.OW = g;
    if (neck2_hl) {
      p.W = neck2_hl, p.W_lo = neck2_hl + (size_t)p.N * p.K;
      p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
      if (planes) p.A_lo = (const half_t*)neck_b + (size_t)Mg * c.out_chans;
      SAMPT_TRY(conv_f16x3(p, s));
    } else {
      SAMPT_TRY(gemm_f32(p, s));
    }
  }
  SAMPT_TRY(layernorm_rows(neck_a, neck3w, neck3b, features, Mg, c.out_chans, 1e-6f, nullptr, 0, ACT_NONE, s));
  return SAMPT_OK;
}

}  // namespace sampt
