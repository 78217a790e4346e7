// SAM prompt encoder + two-way-transformer mask decoder + postprocess, fp32.
//
// `decode` is one SamPredictor.predict_torch pass for a BATCH of F independent frames that share the prompt-token
// count; `track_decode` chains the 1-2 + R passes that SamPt.predict_mask issues per (frame, object)
// (sam_pt/modeling/sam_pt.py:760-837) entirely on the device for all F frames at once:
//   * frame batching: the reference decodes frame after frame; the chains of different frames are independent, so
//     pass r of all frames is ONE launch sequence whose GEMMs have M = F*Nt token rows / F*4096 image rows;
//   * the refinement box comes from an on-device bbox reduction and the reference's `m.sum() < 2 -> break` is a
//     per-frame device-side predicate (later passes still run but are not committed): no host sync, and a launch
//     sequence that depends only on the token count.
#include "engine.h"

namespace sampt {

static void load_attn(const WeightMap& w, const std::string& p, int inner, DecEngine::Attn& a,
                      std::unordered_map<const float*, const half_t*>& w_hl) {
  a.qw = w.f(p + ".q_proj.weight"), a.qb = w.f(p + ".q_proj.bias");
  a.kw = w.f(p + ".k_proj.weight"), a.kb = w.f(p + ".k_proj.bias");
  a.vw = w.f(p + ".v_proj.weight"), a.vb = w.f(p + ".v_proj.bias");
  a.ow = w.f(p + ".out_proj.weight"), a.ob = w.f(p + ".out_proj.bias");
  a.inner = inner;
  const char* names[4] = {".q_proj.weight", ".k_proj.weight", ".v_proj.weight", ".out_proj.weight"};
  const float* ptrs[4] = {a.qw, a.kw, a.vw, a.ow};
  for (int i = 0; i < 4; ++i)
    if (ptrs[i] && w.has(p + names[i] + "_hl")) w_hl[ptrs[i]] = w.h(p + names[i] + "_hl");
}

int DecEngine::init(const WeightMap& w, const DecConfig& cfg) {
  c = cfg;
  if (c.depth > 4) return SAMPT_ERR_UNSUPPORTED;
  const std::string T = "mask_decoder.transformer.";
  for (int i = 0; i < c.depth; ++i) {
    std::string p = T + "layers." + std::to_string(i);
    Layer& L = layer[i];
    load_attn(w, p + ".self_attn", c.C, L.self, w_hl);
    load_attn(w, p + ".cross_attn_token_to_image", c.C / 2, L.t2i, w_hl);
    load_attn(w, p + ".cross_attn_image_to_token", c.C / 2, L.i2t, w_hl);
    L.n1w = w.f(p + ".norm1.weight"), L.n1b = w.f(p + ".norm1.bias");
    L.n2w = w.f(p + ".norm2.weight"), L.n2b = w.f(p + ".norm2.bias");
    L.n3w = w.f(p + ".norm3.weight"), L.n3b = w.f(p + ".norm3.bias");
    L.n4w = w.f(p + ".norm4.weight"), L.n4b = w.f(p + ".norm4.bias");
    L.m1w = w.f(p + ".mlp.lin1.weight"), L.m1b = w.f(p + ".mlp.lin1.bias");
    L.m2w = w.f(p + ".mlp.lin2.weight"), L.m2b = w.f(p + ".mlp.lin2.bias");
  }
  load_attn(w, T + "final_attn_token_to_image", c.C / 2, fin, w_hl);
  auto fused = [&](const std::string& k, int n, FusedProj& f) {
    f.w = w.f(k + "_w"), f.b = w.f(k + "_b"), f.pe = w.f(k + "_pe"), f.n = n;
    if (w.has(k + "_w_hl")) f.hl = w.h(k + "_w_hl");
  };
  for (int i = 0; i < c.depth; ++i) fused(T + "layers." + std::to_string(i) + ".__kvq", 3 * (c.C / 2), kvq[i]);
  fused(T + "__fin_kv", 2 * (c.C / 2), fin_kv);
  nfw = w.f(T + "norm_final_attn.weight"), nfb = w.f(T + "norm_final_attn.bias");
  out_tokens = w.f("mask_decoder.__out_tokens");
  gauss = w.f("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix");
  point_emb = w.f("prompt_encoder.__point_embeddings");
  not_a_point = w.f("prompt_encoder.not_a_point_embed.weight");
  no_mask = w.f("prompt_encoder.no_mask_embed.weight");
  dense_pe = w.f("prompt_encoder.__dense_pe");
  const std::string md = "prompt_encoder.mask_downscaling.";
  me.w0 = w.f(md + "0.weight"), me.b0 = w.f(md + "0.bias"), me.ln0w = w.f(md + "1.weight"), me.ln0b = w.f(md + "1.bias");
  me.w1 = w.f(md + "3.weight"), me.b1 = w.f(md + "3.bias"), me.ln1w = w.f(md + "4.weight"), me.ln1b = w.f(md + "4.bias");
  me.w2 = w.f(md + "6.weight"), me.b2 = w.f(md + "6.bias");
  const std::string U = "mask_decoder.output_upscaling.";
  up0_w = w.f(U + "0.weight_packed"), up0_b = w.f(U + "0.bias");
  upln_w = w.f(U + "1.weight"), upln_b = w.f(U + "1.bias");
  up1_w = w.f(U + "3.weight_packed"), up1_b = w.f(U + "3.bias");
  up0_map = w.i("mask_decoder.__up0_map"), up1_map = w.i("mask_decoder.__up1_map");
  if (w.has(U + "0.weight_packed_hl")) up0_hl = w.h(U + "0.weight_packed_hl");
  if (w.has(U + "3.weight_packed_hl")) up1_hl = w.h(U + "3.weight_packed_hl");
  for (int i = 0; i < 3; ++i) {
    hyp_w[i] = w.f("mask_decoder.output_hypernetworks_mlps.0.layers." + std::to_string(i) + ".weight");
    hyp_b[i] = w.f("mask_decoder.output_hypernetworks_mlps.0.layers." + std::to_string(i) + ".bias");
    for (int m = 0; m < 3; ++m) {
      const std::string hp = "mask_decoder.output_hypernetworks_mlps." + std::to_string(m + 1) + ".layers." + std::to_string(i);
      hypx_w[m][i] = w.f(hp + ".weight"), hypx_b[m][i] = w.f(hp + ".bias");
    }
    iou_w[i] = w.f("mask_decoder.iou_prediction_head.layers." + std::to_string(i) + ".weight");
    iou_b[i] = w.f("mask_decoder.iou_prediction_head.layers." + std::to_string(i) + ".bias");
  }
  if (is_hq()) {
    const std::string M = "mask_decoder.";
    for (int i = 0; i < 3; ++i) {
      hq.mlp_w[i] = w.f(M + "hf_mlp.layers." + std::to_string(i) + ".weight");
      hq.mlp_b[i] = w.f(M + "hf_mlp.layers." + std::to_string(i) + ".bias");
    }
    auto seq = [&](const std::string& p, const float*& w0, const float*& b0, const float*& lw, const float*& lb,
                   const float*& w1, const float*& b1) {
      w0 = w.f(M + p + ".0.weight_packed"), b0 = w.f(M + p + ".0.bias");
      lw = w.f(M + p + ".1.weight"), lb = w.f(M + p + ".1.bias");
      w1 = w.f(M + p + ".3.weight_packed"), b1 = w.f(M + p + ".3.bias");
    };
    seq("compress_vit_feat", hq.cv0_w, hq.cv0_b, hq.cvln_w, hq.cvln_b, hq.cv1_w, hq.cv1_b);
    seq("embedding_encoder", hq.ee0_w, hq.ee0_b, hq.eeln_w, hq.eeln_b, hq.ee1_w, hq.ee1_b);
    seq("embedding_maskfeature", hq.mf0_w, hq.mf0_b, hq.mfln_w, hq.mfln_b, hq.mf1_w, hq.mf1_b);
    auto opt = [&](const std::string& k) { return w.has(M + k) ? w.h(M + k) : nullptr; };
    hq.cv0_hl = opt("compress_vit_feat.0.weight_packed_hl"), hq.cv1_hl = opt("compress_vit_feat.3.weight_packed_hl");
    hq.ee0_hl = opt("embedding_encoder.0.weight_packed_hl"), hq.ee1_hl = opt("embedding_encoder.3.weight_packed_hl");
    if (w.has(M + "embedding_maskfeature.0.weight_packed_hl")) hq.mf0_hl = w.h(M + "embedding_maskfeature.0.weight_packed_hl");
    if (w.has(M + "embedding_maskfeature.3.weight_packed_hl")) hq.mf1_hl = w.h(M + "embedding_maskfeature.3.weight_packed_hl");
  }
  if (!w.missing.empty()) {
    error = "DecEngine: missing weights: " + w.missing;
    return SAMPT_ERR_ARG;
  }
  return SAMPT_OK;
}

namespace {
struct L {
  hipStream_t s;
  float* skws;
  size_t skn;
  const std::unordered_map<const float*, const half_t*>* w_hl = nullptr;
  int lin(const float* A, int M, int K, const float* W, const float* b, float* C, int N, int act = ACT_NONE,
          const float* res = nullptr, int lda = 0) const {
    GemmP p;
    p.A = A, p.W = W, p.bias = b, p.C = C, p.res = res;
    p.M = M, p.N = N, p.K = K, p.lda = lda ? lda : K, p.ldw = K, p.ldc = N, p.ldr = N, p.act = act;
    p.splitk_ws = skws, p.splitk_ws_floats = skn;
    if (w_hl && M >= 2048 && !lda && K % 32 == 0 && N % 4 == 0) {     // image-token projections, split-fp16 planes packed
      auto it = w_hl->find(W);
      if (it != w_hl->end()) {
        p.W = it->second, p.W_lo = it->second + (size_t)N * K;
        p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
        p.conv = 1, p.cH = M, p.cW = 1, p.cC = K, p.KH = 1, p.KW = 1, p.cstride = 1, p.cpad = 0, p.OH = M, p.OW = 1;
        p.splitk_ws = nullptr, p.splitk_ws_floats = 0;
        return conv_f16x3(p, s);
      }
    }
    return gemm_f32(p, s);
  }
};

struct Bufs {
  float *tokens, *qin, *keys, *Q, *K, *V, *att, *hid, *up0, *up1, *t0, *t1, *t2, *me0, *me1;
  float* part = nullptr;      // split-key partial states of the token -> image attention (attn_t2i)
  size_t part_floats = 0;
  float* kvq = nullptr;       // fused image-side projections [F*P][K | V | Q']
};
}  // namespace

// fused image-side projection: out [M][f.n] = A W^T + b + pe[row % P]   (see DecEngine::FusedProj)
static int fused_proj(const L& l, const DecEngine::FusedProj& f, const float* A, long M, int C, int P, float* out) {
  GemmP p;
  p.A = A, p.bias = f.b, p.C = out, p.res = f.pe, p.res_mod = P;
  p.M = (int)M, p.N = f.n, p.K = C, p.lda = C, p.ldw = C, p.ldc = f.n, p.ldr = f.n;
  if (f.hl && C % 32 == 0) {
    p.W = f.hl, p.W_lo = f.hl + (size_t)f.n * C, p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
    p.conv = 1, p.cH = (int)M, p.cW = 1, p.cC = C, p.KH = 1, p.KW = 1, p.cstride = 1, p.cpad = 0, p.OH = (int)M, p.OW = 1;
    return conv_f16x3(p, l.s);
  }
  p.W = f.w;
  return gemm_f32(p, l.s);
}

// tail of an attention block: out = LN(resid + out_proj(att))
static int attn_tail(const L& l, const DecEngine::Attn& a, int C, long rows, const float* att, const float* resid, float* out,
                     const float* lnw, const float* lnb, hipStream_t s) {
  if (l.w_hl && resid && g_gemm_x3_wres && g_gemm_x3_epi) {      // image-token side at scale: projection + residual + LayerNorm in one kernel
    auto it = l.w_hl->find(a.ow);
    if (it != l.w_hl->end()) {
      GemmP p;
      p.A = att, p.W = it->second, p.W_lo = it->second + (size_t)C * a.inner, p.bias = a.ob, p.C = out, p.res = resid;
      p.M = (int)rows, p.N = C, p.K = a.inner, p.ldw = a.inner, p.ldc = C, p.ldr = C;
      p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
      p.conv = 1, p.cH = (int)rows, p.cW = 1, p.cC = a.inner, p.KH = 1, p.KW = 1, p.cstride = 1, p.cpad = 0, p.OH = (int)rows, p.OW = 1;
      p.epi = 3, p.epi_a = lnw, p.epi_b = lnb, p.epi_eps = 1e-5f;
      if (gemm_x3_wres_ln_eligible(p)) return gemm_x3_wres(p, s);
    }
  }
  SAMPT_TRY(l.lin(att, (int)rows, a.inner, a.ow, a.ob, out, C, ACT_NONE, resid));
  return layernorm_rows(out, lnw, lnb, out, rows, C, 1e-5f, nullptr, 0, ACT_NONE, s);
}

// attention block over F frames: out = LN(resid + out_proj(attn(q_in Wq, k_in Wk, v_in Wv)))  (resid null: replace)
static int attn_block(const L& l, const DecEngine::Attn& a, int heads, int C, int F, const float* q_in, int Nq,
                      const float* k_in, const float* v_in, int Nk, bool few_keys, const int* nk_item, Bufs& b,
                      const float* resid, float* out, const float* lnw, const float* lnb, hipStream_t s) {
  SAMPT_TRY(l.lin(q_in, F * Nq, C, a.qw, a.qb, b.Q, a.inner));
  SAMPT_TRY(l.lin(k_in, F * Nk, C, a.kw, a.kb, b.K, a.inner));
  SAMPT_TRY(l.lin(v_in, F * Nk, C, a.vw, a.vb, b.V, a.inner));
  const int hd = a.inner / heads;
  if (few_keys) SAMPT_TRY(attn_fewkeys(b.Q, b.K, b.V, b.att, F, Nq, Nk, heads, hd, nk_item, s));
  else if (hd == 16 && heads == 8 && Nk > 256 && !nk_item)     // token -> image: the K / V stream decides, read it once
    SAMPT_TRY(attn_t2i(b.Q, b.K, b.V, b.att, F, Nq, Nk, b.part, b.part_floats, s));
  else SAMPT_TRY(attn_rowblock(b.Q, b.K, b.V, b.att, F, Nq, Nk, heads, hd, nk_item, s));
  SAMPT_TRY(l.lin(b.att, F * Nq, a.inner, a.ow, a.ob, out, C, ACT_NONE, resid));
  return layernorm_rows(out, lnw, lnb, out, (long)F * Nq, C, 1e-5f, nullptr, 0, ACT_NONE, s);
}

// two ConvT2x2s2 stages with LayerNorm2d + GELU in between (pixel shuffle through the row maps); the second stage
// optionally applies `act` and adds `res` (same layout as out):  x [F*P][K0] -> mid [F*4P][N0] -> out [F*16P][N1]
// dot_hyper non-null (plain SAM, single mask): the second stage's output is never materialised — its rows are dotted with the frame's
// hypernetwork vector dot_hyper[f * dot_ld + 0 .. N1) in the GEMM's epilogue and land in low_out [F][16 P] (gemm_x3_wres.hip, epi = 2);
// returns through *fused_dot whether that happened (the caller runs sam_mask_dot otherwise)
static int convt_pair(const DecEngine& e, int F, const float* x, int K0, const float* w0, const float* b0, int N0,
                      const float* lnw, const float* lnb, const float* w1, const float* b1, int N1, int act,
                      const float* res, float* mid, float* out, float* skws, size_t skn, hipStream_t s,
                      const half_t* w0_hl = nullptr, const half_t* w1_hl = nullptr, const float* dot_hyper = nullptr, int dot_ld = 0,
                      float* low_out = nullptr, bool* fused_dot = nullptr) {
  const long FP = (long)F * e.c.grid * e.c.grid, P = (long)e.c.grid * e.c.grid;
  if (w0_hl && w1_hl && K0 % 32 == 0 && N0 % 32 == 0 && N1 % 4 == 0) {
    // Each stage as ONE 3-term split-fp16 GEMM over all four (dy, dx) sub-pixels (N = 4 * cout): the input is read once
    // instead of four times and the pixel shuffle is address arithmetic in the epilogue (conv_f16x3.hip, GemmP::shuf_g).
    auto stage = [&](const float* A, long M, int K, const half_t* hl, const float* bias, int nsub, int g, float* C, int a,
                     const float* r) {
      GemmP p;
      p.A = A, p.W = hl, p.W_lo = hl + (size_t)4 * nsub * K, p.bias = bias, p.C = C, p.res = r, p.act = a;
      p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
      p.M = (int)M, p.N = 4 * nsub, p.K = K, p.ldw = K, p.ldc = nsub, p.ldr = nsub;
      p.conv = 1, p.cH = (int)M, p.cW = 1, p.cC = K, p.KH = 1, p.KW = 1, p.cstride = 1, p.cpad = 0, p.OH = (int)M, p.OW = 1;
      p.shuf_g = g, p.shuf_n = nsub;
      return conv_f16x3(p, s);
    };
    // Where the weights-resident kernel takes the launch (gemm_x3_wres.hip: F * P >= 16384 rows, the decoder's own channel counts)
    // the LayerNorm2d + GELU between the stages runs in the first stage's epilogue, and (dot_hyper) the mask's dot product in the
    // second's — the same arithmetic as the kernels they replace, operation for operation (tests/test_gpu_kernels.py)
    auto fused = [&](const float* A, long M, int K, const half_t* hl, const float* bias, int nsub, int g, float* C, int epi) {
      GemmP p;
      p.A = A, p.W = hl, p.W_lo = hl + (size_t)4 * nsub * K, p.bias = bias, p.C = C, p.act = ACT_GELU;
      p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
      p.M = (int)M, p.N = 4 * nsub, p.K = K, p.ldw = K, p.ldc = epi == 2 ? 1 : nsub, p.ldr = nsub;
      p.conv = 1, p.cH = (int)M, p.cW = 1, p.cC = K, p.KH = 1, p.KW = 1, p.cstride = 1, p.cpad = 0, p.OH = (int)M, p.OW = 1;
      p.shuf_g = g, p.shuf_n = nsub, p.epi = epi;
      if (epi == 1) p.epi_a = lnw, p.epi_b = lnb, p.epi_eps = 1e-6f;
      else p.epi_a = dot_hyper, p.epi_ld = dot_ld;
      return p;
    };
    const GemmP f0 = fused(x, FP, K0, w0_hl, b0, N0, e.c.grid, mid, 1);
    if (g_gemm_x3_wres && g_gemm_x3_epi && gemm_x3_wres_eligible(f0)) {
      SAMPT_TRY(gemm_x3_wres(f0, s));
    } else {
      SAMPT_TRY(stage(x, FP, K0, w0_hl, b0, N0, e.c.grid, mid, ACT_NONE, nullptr));
      SAMPT_TRY(layernorm_rows(mid, lnw, lnb, mid, 4L * FP, N0, 1e-6f, nullptr, 0, ACT_GELU, s));
    }
    if (dot_hyper && act == ACT_GELU && !res) {
      const GemmP f1 = fused(mid, 4 * FP, N0, w1_hl, b1, N1, 2 * e.c.grid, low_out, 2);
      if (g_gemm_x3_wres && g_gemm_x3_epi && gemm_x3_wres_eligible(f1)) {
        if (fused_dot) *fused_dot = true;
        return gemm_x3_wres(f1, s);
      }
    }
    return stage(mid, 4 * FP, N0, w1_hl, b1, N1, 2 * e.c.grid, out, act, res);
  }
  GemmP p;
  p.A = x, p.W = w0, p.bias = b0, p.C = mid, p.rowmap = e.up0_map;
  p.M = (int)FP, p.N = N0, p.K = K0, p.lda = K0, p.ldw = K0, p.ldc = N0;
  p.nb1 = 4, p.sW1 = (long)N0 * K0, p.sRowmap1 = (long)e.max_frames * P;
  (void)skws, (void)skn;
  SAMPT_TRY(gemm_f32(p, s));
  SAMPT_TRY(layernorm_rows(mid, lnw, lnb, mid, 4L * FP, N0, 1e-6f, nullptr, 0, ACT_GELU, s));
  GemmP q;
  q.A = mid, q.W = w1, q.bias = b1, q.C = out, q.rowmap = e.up1_map, q.res = res;
  q.M = (int)(4 * FP), q.N = N1, q.K = N0, q.lda = N0, q.ldw = N0, q.ldc = N1, q.ldr = N1, q.act = act;
  q.nb1 = 4, q.sW1 = (long)N1 * N0, q.sRowmap1 = 4L * e.max_frames * P;
  return gemm_f32(q, s);
}

int DecEngine::hq_features(int F, const float* features, const float* interm, float* out, Arena& ws, hipStream_t s) {
  if (!is_hq() || F <= 0 || F > max_frames) return SAMPT_ERR_ARG;
  const int C = c.C;
  const size_t FP = (size_t)F * c.grid * c.grid;
  float* mid = ws.f32(4 * FP * C);
  if (!ws.ok()) return SAMPT_ERR_WORKSPACE;
  if (ws.dry()) return SAMPT_OK;
  // out = compress_vit_feat(interm) ; out += embedding_encoder(features)
  SAMPT_TRY(convt_pair(*this, F, interm, c.vit_dim, hq.cv0_w, hq.cv0_b, C, hq.cvln_w, hq.cvln_b, hq.cv1_w, hq.cv1_b, C / 8,
                       ACT_NONE, nullptr, mid, out, nullptr, 0, s, hq.cv0_hl, hq.cv1_hl));
  return convt_pair(*this, F, features, C, hq.ee0_w, hq.ee0_b, C / 4, hq.eeln_w, hq.eeln_b, hq.ee1_w, hq.ee1_b, C / 8,
                    ACT_NONE, out, mid, out, nullptr, 0, s, hq.ee0_hl, hq.ee1_hl);
}

int DecEngine::decode(int F, const float* features, const float* hq_feat, const float* pts, const int* labels, int k,
                      const int* k_item, int ld_pts, const float* box, const float* mask_in, int in_h, int in_w, int oh, int ow,
                      float* logits_out, float* iou_out, float* low_out, int* bbox_out, Arena& ws, hipStream_t s) {
  const int g = c.grid, P = g * g, C = c.C, H = c.heads, NO = n_out();
  const int nsparse = k + (box ? 2 : 1), Nt = NO + nsparse;
  if (Nt > 4096 || k < 0 || F <= 0 || F > max_frames) return SAMPT_ERR_UNSUPPORTED;   // attn_rowblock: <= 4096 keys
  if (is_hq() != (hq_feat != nullptr)) return SAMPT_ERR_ARG;
  const size_t FP = (size_t)F * P, FT = (size_t)F * Nt;
  Bufs b;
  b.tokens = ws.f32(FT * C);
  float* queries = ws.f32(FT * C);
  b.qin = ws.f32(FT * C);
  b.kvq = ws.f32(FP * 3 * (C / 2));
  b.keys = ws.f32(FP * C);
  const size_t Fmx = FP > FT ? FP : FT;   // projections run on image tokens AND on prompt tokens (Nt may exceed g*g)
  b.Q = ws.f32(Fmx * C);
  b.K = ws.f32(Fmx * C);
  b.V = ws.f32(Fmx * C);
  b.att = ws.f32(Fmx * C);
  b.hid = ws.f32(FT * c.mlp);
  b.part_floats = attn_t2i_workspace_floats(F, Nt, P);
  b.part = ws.f32(b.part_floats ? b.part_floats : 4);
  b.up0 = ws.f32(4 * FP * (C / 4));
  b.up1 = ws.f32(16 * FP * (C / 8));
  b.t0 = ws.f32((size_t)F * C), b.t1 = ws.f32((size_t)F * C), b.t2 = ws.f32((size_t)F * C);
  b.me0 = ws.f32(4 * FP * 4), b.me1 = ws.f32(FP * 16);
  float *uh0 = nullptr, *uh1 = nullptr, *t3 = nullptr;
  if (is_hq()) uh0 = ws.f32(16 * FP * (C / 4)), uh1 = ws.f32(16 * FP * (C / 8)), t3 = ws.f32((size_t)F * C);
  int* bbox_partial = (int*)ws.get((size_t)F * bbox_partial_ints(oh, ow) * sizeof(int));
  int* ntok = (int*)ws.get((size_t)F * sizeof(int));           // valid tokens per item (ragged batch)
  float* iou4 = ws.f32((size_t)F * 4);
  if (multimask && (is_hq() || F != 1)) return SAMPT_ERR_UNSUPPORTED;
  const size_t skn = (size_t)16 * FT * C;
  float* skws = ws.f32(skn);
  if (!ws.ok()) return SAMPT_ERR_WORKSPACE;
  if (ws.dry()) return SAMPT_OK;
  L l{s, skws, skn, w_hl.empty() ? nullptr : &w_hl};

  // ---- prompt encoder
  SAMPT_TRY(sam_tokens(out_tokens, NO, pts, labels, k, ld_pts, box, gauss, point_emb, not_a_point, (float)c.img, F,
                       k_item, ntok, b.tokens, s));
  const int* nkt = k_item ? ntok : nullptr;                    // tokens-as-keys mask; null = all Nt rows are valid
  if (mask_in) SAMPT_TRY(sam_mask_embed_src(mask_in, g, F, me, features, b.me0, b.me1, b.keys, s));
  else SAMPT_TRY(add_bcast(features, no_mask, b.keys, (long)FP * C, C, s));

  // ---- two-way transformer: query PE = the prompt tokens, key PE = dense positional encoding
  const float* qpe = b.tokens;
  const long nT = (long)FT * C;
  const int inner = C / 2;
  if (inner != 16 * H || inner != 128) return SAMPT_ERR_UNSUPPORTED;   // attn_t2i / attn_fewkeys: 8 heads x 16 channels
  for (int i = 0; i < c.depth; ++i) {
    const Layer& Ly = layer[i];
    if (i == 0) {
      SAMPT_TRY(attn_block(l, Ly.self, H, C, F, b.tokens, Nt, b.tokens, b.tokens, Nt, false, nkt, b, nullptr, queries,
                           Ly.n1w, Ly.n1b, s));
    } else {
      SAMPT_TRY(add_bcast(queries, qpe, b.qin, nT, nT, s));
      SAMPT_TRY(attn_block(l, Ly.self, H, C, F, b.qin, Nt, b.qin, queries, Nt, false, nkt, b, queries, queries, Ly.n1w,
                           Ly.n1b, s));
    }
    // tokens -> image.  K = (keys + pe) Wk, V = keys Wv and the image -> token query (keys + pe) Wq' below all read the
    // same image tokens: one fused projection, kvq [F*P][K | V | Q'], the positional terms folded into its epilogue
    SAMPT_TRY(add_bcast(queries, qpe, b.qin, nT, nT, s));
    SAMPT_TRY(fused_proj(l, kvq[i], b.keys, (long)FP, C, P, b.kvq));
    SAMPT_TRY(l.lin(b.qin, (int)FT, C, Ly.t2i.qw, Ly.t2i.qb, b.Q, inner));
    SAMPT_TRY(attn_t2i(b.Q, b.kvq, b.kvq + inner, b.att, F, Nt, P, b.part, b.part_floats, s, 3 * inner));
    SAMPT_TRY(attn_tail(l, Ly.t2i, C, (long)FT, b.att, queries, queries, Ly.n2w, Ly.n2b, s));
    // MLP (ReLU)
    SAMPT_TRY(l.lin(queries, (int)FT, C, Ly.m1w, Ly.m1b, b.hid, c.mlp, ACT_RELU));
    SAMPT_TRY(l.lin(b.hid, (int)FT, c.mlp, Ly.m2w, Ly.m2b, queries, C, ACT_NONE, queries));
    SAMPT_TRY(layernorm_rows(queries, Ly.n3w, Ly.n3b, queries, (long)FT, C, 1e-5f, nullptr, 0, ACT_NONE, s));
    // image -> tokens (queries: the Q' slice of kvq)
    SAMPT_TRY(add_bcast(queries, qpe, b.qin, nT, nT, s));
    SAMPT_TRY(l.lin(b.qin, (int)FT, C, Ly.i2t.kw, Ly.i2t.kb, b.K, inner));
    SAMPT_TRY(l.lin(queries, (int)FT, C, Ly.i2t.vw, Ly.i2t.vb, b.V, inner));
    SAMPT_TRY(attn_fewkeys(b.kvq + 2 * inner, b.K, b.V, b.att, F, P, Nt, H, inner / H, nkt, s, 3 * inner));
    SAMPT_TRY(attn_tail(l, Ly.i2t, C, (long)FP, b.att, b.keys, b.keys, Ly.n4w, Ly.n4b, s));
  }
  SAMPT_TRY(add_bcast(queries, qpe, b.qin, nT, nT, s));
  SAMPT_TRY(fused_proj(l, fin_kv, b.keys, (long)FP, C, P, b.kvq));
  SAMPT_TRY(l.lin(b.qin, (int)FT, C, fin.qw, fin.qb, b.Q, inner));
  SAMPT_TRY(attn_t2i(b.Q, b.kvq, b.kvq + inner, b.att, F, Nt, P, b.part, b.part_floats, s, 2 * inner));
  SAMPT_TRY(attn_tail(l, fin, C, (long)FT, b.att, queries, queries, nfw, nfb, s));

  // ---- upscaling: ConvT2x2s2 (C -> C/4) + LN2d + GELU ; ConvT2x2s2 (C/4 -> C/8) + GELU   (pixel shuffle via row maps
  //      that cover max_frames frames: map[(dy,dx)][f*P + p] = f*4P + (2y+dy)*2g + 2x+dx)
  // (plain SAM, one mask per item: the hypernetwork vector of mask token 0 first — it only needs the tokens — so that the second
  //  transposed convolution can dot its rows with it instead of writing them)
  bool dot_done = false;
  const bool dot_early = !multimask && !is_hq();
  if (dot_early) {
    const float* mask_tok0 = queries + 1 * C;
    SAMPT_TRY(l.lin(mask_tok0, F, C, hyp_w[0], hyp_b[0], b.t0, C, ACT_RELU, nullptr, Nt * C));
    SAMPT_TRY(l.lin(b.t0, F, C, hyp_w[1], hyp_b[1], b.t1, C, ACT_RELU));
    SAMPT_TRY(l.lin(b.t1, F, C, hyp_w[2], hyp_b[2], b.t2, C / 8, ACT_NONE));
  }
  SAMPT_TRY(convt_pair(*this, F, b.keys, C, up0_w, up0_b, C / 4, upln_w, upln_b, up1_w, up1_b, C / 8, ACT_GELU, nullptr,
                       b.up0, b.up1, nullptr, 0, s, up0_hl, up1_hl, dot_early ? b.t2 : nullptr, C / 8, low_out, &dot_done));
  if (multimask) {
    // multimask_output=True (MaskDecoder.forward: mask_slice = slice(1, None)): masks and IoUs of mask tokens 1..3;
    // low_out [3][4g][4g], logits_out [3][oh][ow], iou_out [3]
    for (int m = 0; m < 3; ++m) {
      SAMPT_TRY(l.lin(queries + (2 + m) * C, 1, C, hypx_w[m][0], hypx_b[m][0], b.t0, C, ACT_RELU, nullptr, Nt * C));
      SAMPT_TRY(l.lin(b.t0, 1, C, hypx_w[m][1], hypx_b[m][1], b.t1, C, ACT_RELU));
      SAMPT_TRY(l.lin(b.t1, 1, C, hypx_w[m][2], hypx_b[m][2], b.t2, C / 8, ACT_NONE));
      SAMPT_TRY(sam_mask_dot(b.up1, b.t2, C / 8, nullptr, nullptr, 0, low_out + (size_t)m * 16 * P, 1, 16 * P, C / 8, s));
    }
    SAMPT_TRY(l.lin(queries, 1, C, iou_w[0], iou_b[0], b.t0, C, ACT_RELU, nullptr, Nt * C));
    SAMPT_TRY(l.lin(b.t0, 1, C, iou_w[1], iou_b[1], b.t1, C, ACT_RELU));
    SAMPT_TRY(l.lin(b.t1, 1, C, iou_w[2], iou_b[2], iou4, 4, ACT_NONE));
    if (hipMemcpyAsync(iou_out, iou4 + 1, 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return SAMPT_ERR_HIP;
    return sam_postprocess_bbox(low_out, 4 * g, c.img, in_h, in_w, logits_out, oh, ow, 3, nullptr, nullptr, s);
  }
  // ---- hypernetwork MLP of mask token 0 (multimask_output=False keeps slice 0 only) and the IoU head;
  //      A = row 1 (mask token 0) / row 0 (iou token) of every frame's token matrix: lda = Nt*C
  const float* mask_tok = queries + 1 * C;
  if (!dot_early) {
    SAMPT_TRY(l.lin(mask_tok, F, C, hyp_w[0], hyp_b[0], b.t0, C, ACT_RELU, nullptr, Nt * C));
    SAMPT_TRY(l.lin(b.t0, F, C, hyp_w[1], hyp_b[1], b.t1, C, ACT_RELU));
    SAMPT_TRY(l.lin(b.t1, F, C, hyp_w[2], hyp_b[2], b.t2, C / 8, ACT_NONE));
  }
  if (!is_hq()) {
    if (!dot_done) SAMPT_TRY(sam_mask_dot(b.up1, b.t2, C / 8, nullptr, nullptr, 0, low_out, F, 16 * P, C / 8, s));
  } else {
    // HQ-SAM: upscaled_hq = conv3x3(GELU(LN2d(conv3x3(upscaled)))) + hq_features ;  mask = <hyper0, upscaled> +
    // <hf_mlp(hq token), upscaled_hq>   (MaskDecoderHQ.predict_masks, hq_token_only=False)
    const int Lr = 4 * g;
    GemmP p;
    p.A = b.up1, p.W = hq.mf0_w, p.bias = hq.mf0_b, p.C = uh0;
    p.M = (int)(16 * FP), p.N = C / 4, p.K = 9 * (C / 8), p.ldw = p.K, p.ldc = C / 4;
    p.conv = 1, p.cH = Lr, p.cW = Lr, p.cC = C / 8, p.KH = 3, p.KW = 3, p.cstride = 1, p.cpad = 1, p.OH = Lr, p.OW = Lr;
    if (hq.mf0_hl) {   // 3-term split-fp16 MFMAs: fp32-grade, 2.25x the f32 MFMA rate (the two convs are 4.8 of an HQ pass's 8.6 GFLOP)
      p.W = hq.mf0_hl, p.W_lo = hq.mf0_hl + (size_t)p.N * p.K, p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
      SAMPT_TRY(conv_f16x3(p, s));
    } else {
      SAMPT_TRY(gemm_f32(p, s));
    }
    SAMPT_TRY(layernorm_rows(uh0, hq.mfln_w, hq.mfln_b, uh0, 16L * FP, C / 4, 1e-6f, nullptr, 0, ACT_GELU, s));
    GemmP q;
    q.A = uh0, q.W = hq.mf1_w, q.bias = hq.mf1_b, q.C = uh1, q.res = hq_feat;
    q.M = (int)(16 * FP), q.N = C / 8, q.K = 9 * (C / 4), q.ldw = q.K, q.ldc = C / 8, q.ldr = C / 8;
    q.conv = 1, q.cH = Lr, q.cW = Lr, q.cC = C / 4, q.KH = 3, q.KW = 3, q.cstride = 1, q.cpad = 1, q.OH = Lr, q.OW = Lr;
    if (hq.mf1_hl) {
      q.W = hq.mf1_hl, q.W_lo = hq.mf1_hl + (size_t)q.N * q.K, q.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
      SAMPT_TRY(conv_f16x3(q, s));
    } else {
      SAMPT_TRY(gemm_f32(q, s));
    }
    const float* hq_tok = queries + 5 * C;
    SAMPT_TRY(l.lin(hq_tok, F, C, hq.mlp_w[0], hq.mlp_b[0], b.t0, C, ACT_RELU, nullptr, Nt * C));
    SAMPT_TRY(l.lin(b.t0, F, C, hq.mlp_w[1], hq.mlp_b[1], b.t1, C, ACT_RELU));
    SAMPT_TRY(l.lin(b.t1, F, C, hq.mlp_w[2], hq.mlp_b[2], t3, C / 8, ACT_NONE));
    SAMPT_TRY(sam_mask_dot(b.up1, b.t2, C / 8, uh1, t3, C / 8, low_out, F, 16 * P, C / 8, s));
  }
  SAMPT_TRY(l.lin(queries, F, C, iou_w[0], iou_b[0], b.t0, C, ACT_RELU, nullptr, Nt * C));
  SAMPT_TRY(l.lin(b.t0, F, C, iou_w[1], iou_b[1], b.t1, C, ACT_RELU));
  SAMPT_TRY(l.lin(b.t1, F, C, iou_w[2], iou_b[2], iou_out, 1, ACT_NONE));  // N = 1: only IoU slot 0 is needed
  // ---- Sam.postprocess_masks (+ bbox of logits > 0)
  SAMPT_TRY(sam_postprocess_bbox(low_out, 4 * g, c.img, in_h, in_w, logits_out, oh, ow, F, bbox_out, bbox_partial, s));
  return SAMPT_OK;
}

int DecEngine::track_decode(int F, const float* features, const float* hq_feat, const float* pts, const int* labels,
                            int k, const int* k_item, const int* npos_item, int ld_pts, int n_pos_first, int R, float iou_thr, int in_h, int in_w, int oh, int ow,
                            float* final_logits, float* score_out, Arena& ws, hipStream_t s) {
  const int g = c.grid, Lr = 4 * g;
  const long nlog = (long)oh * ow, nlow = (long)Lr * Lr;
  float* cur_logits = ws.f32((size_t)F * nlog);
  float* cand_logits = ws.f32((size_t)F * nlog);
  float* cur_low = ws.f32((size_t)F * nlow);
  float* cand_low = ws.f32((size_t)F * nlow);
  float* low0 = ws.f32((size_t)F * nlow);
  float* cur_iou = ws.f32(F);
  float* cand_iou = ws.f32(F);
  float* boxf = ws.f32((size_t)F * 4);
  int* cur_bb = (int*)ws.get((size_t)F * 5 * sizeof(int));
  int* cand_bb = (int*)ws.get((size_t)F * 5 * sizeof(int));
  int* active = (int*)ws.get((size_t)F * sizeof(int));
  size_t mark = ws.off;
  if (ws.dry()) {  // measure the per-pass scratch once (all passes reuse it); box + mask = the largest variant
    SAMPT_TRY(decode(F, features, hq_feat, pts, labels, k, nullptr, ld_pts, pts, cur_low, in_h, in_w, oh, ow, cur_logits,
                     cur_iou, cur_low, cur_bb, ws, s));
    return SAMPT_OK;
  }
  if (!ws.ok()) return SAMPT_ERR_WORKSPACE;
  const float* mask_in = nullptr;
  if (n_pos_first >= 0) {  // negative_points_per_mask > 0 (sam_pt.py:791-807): positives only, then all + low-res mask
    ws.off = mark;
    SAMPT_TRY(decode(F, features, hq_feat, pts, labels, n_pos_first, npos_item, ld_pts, nullptr, nullptr, in_h, in_w, oh, ow,
                     cand_logits, cand_iou, low0, nullptr, ws, s));
    mask_in = low0;
  }
  ws.off = mark;
  SAMPT_TRY(decode(F, features, hq_feat, pts, labels, k, k_item, ld_pts, nullptr, mask_in, in_h, in_w, oh, ow, cur_logits,
                   cur_iou, cur_low, cur_bb, ws, s));
  if (R > 0) {
    for (int r = 0; r < R; ++r) {
      SAMPT_TRY(sam_refine_gate(active, cur_bb, boxf, F, r == 0, s));
      ws.off = mark;
      SAMPT_TRY(decode(F, features, hq_feat, pts, labels, k, k_item, ld_pts, boxf, cur_low, in_h, in_w, oh, ow, cand_logits,
                       cand_iou, cand_low, cand_bb, ws, s));
      SAMPT_TRY(sam_commit(active, cand_logits, cur_logits, nlog, cand_low, cur_low, nlow, cand_iou, cur_iou, cand_bb,
                           cur_bb, F, s));
    }
  }
  return sam_finalize_mask(cur_logits, cur_iou, iou_thr, final_logits, score_out, nlog, F, s);
}

}  // namespace sampt
