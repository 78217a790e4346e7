// PIPS++ engine (row f4): whole-chunk iterative refinement (pips_plus_plus.py:436-546) on top of the PIPS kernels — the
// feature pyramid, the fused local correlation sampler (three templates per iteration) and the bilinear feature gather
// are shared; the DeltaBlock (1-D ResNet over time, :263-342) runs as implicit-GEMM convolutions over [n][S][1][C].
#include "engine.h"

namespace sampt {

static const int kBlocks[8][2] = {{128, 128}, {128, 128}, {128, 256}, {256, 256}, {256, 512}, {512, 512}, {512, 1024},
                                  {1024, 1024}};

int Pips2Engine::init(const WeightMap& w, int stride_) {
  stride = stride_;
  enc.stride = stride_;
  int rc = enc.init_fnet(w);
  if (rc != SAMPT_OK) {
    error = enc.error;
    return rc;
  }
  auto conv = [&](const std::string& p, int cin, int cout, Conv1& c) {
    c.w = w.f(p + ".conv.weight"), c.b = w.f(p + ".conv.bias"), c.cin = cin, c.cout = cout;
  };
  conv("delta_block.first_block_conv", 720, 128, first);            // 718 input channels zero-padded to 720
  for (int i = 0; i < 8; ++i) {
    const std::string p = "delta_block.basicblock_list." + std::to_string(i);
    conv(p + ".conv1", kBlocks[i][0], kBlocks[i][1], blk[i][0]);
    conv(p + ".conv2", kBlocks[i][1], kBlocks[i][1], blk[i][1]);
  }
  dense_w = w.f("delta_block.dense.weight"), dense_b = w.f("delta_block.dense.bias");
  omega = w.f("__omega");
  if (!w.missing.empty()) {
    error = "Pips2Engine: missing weights: " + w.missing;
    return SAMPT_ERR_ARG;
  }
  return SAMPT_OK;
}

// Conv1dPad(k = 3, "same") over the S frames of every point: y[pt*S+s][co] = b + sum_{d,ci} x[pt*S+s+d-1][ci] W[co][d][ci]
static int conv1d(const Pips2Engine::Conv1& c, const float* x, int n, int S, float* y, int act, const float* res,
                  hipStream_t s) {
  GemmP p;
  p.A = x, p.W = c.w, p.bias = c.b, p.C = y, p.res = res;
  p.M = n * S, p.N = c.cout, p.K = 3 * c.cin, p.ldw = p.K, p.ldc = c.cout, p.ldr = c.cout, p.act = act;
  p.conv = 1, p.cH = S, p.cW = 1, p.cC = c.cin, p.KH = 3, p.KW = 1, p.cstride = 1, p.cpad = 1, p.cpadw = 0;
  p.OH = S, p.OW = 1;
  return gemm_f32(p, s);
}

int Pips2Engine::update(const PyramidLevels& pyr, const int* frame_idx, int n, int S, const float* trajs0, int have_init,
                        float* const feats[3], int iters, float* trajs_out, Arena& ws, hipStream_t s) {
  if (n <= 0 || S <= 0 || iters <= 0) return SAMPT_ERR_ARG;
  const size_t R = (size_t)n * S;
  const int LDX = 720;
  float* coords = ws.f32((size_t)S * n * 2);
  float* bak = ws.f32((size_t)n * 2);
  float* x = ws.f32(R * LDX);
  float* buf[4];
  for (int i = 0; i < 4; ++i) buf[i] = ws.f32(R * 1024);
  float* delta = ws.f32(R * 2);
  if (!ws.ok()) return SAMPT_ERR_WORKSPACE;
  if (ws.dry()) return SAMPT_OK;
  const int H = pyr.H[0], W = pyr.W[0];
  SAMPT_TRY(pips2_init(trajs0, pyr.base[0], H, W, frame_idx, (float)stride, S, n, have_init, coords, bak, feats[0], feats[1],
                       feats[2], s));
  for (int it = 0; it < iters; ++it) {
    if (it >= 1) SAMPT_TRY(pips2_templates(pyr.base[0], H, W, frame_idx, coords, S, n, feats[1], feats[2], s));
    for (int t = 0; t < 3; ++t) SAMPT_TRY(pips_corr_sample(pyr, frame_idx, S, n, 128, feats[t], coords, x, LDX, 196 * t, s));
    SAMPT_TRY(pips2_build_input(coords, omega, S, n, x, LDX, s));
    // ---- DeltaBlock: h = relu(conv(x)); 8 residual blocks; relu; dense
    float *h = buf[0], *a = buf[1], *b = buf[2], *c = buf[3];
    SAMPT_TRY(conv1d(first, x, n, S, h, ACT_RELU, nullptr, s));
    for (int i = 0; i < 8; ++i) {
      const int cin = kBlocks[i][0], cout = kBlocks[i][1];
      const float* in1 = h;
      if (i > 0) {                                   // norm1 + relu1 (skipped in the first block, :80-86)
        SAMPT_TRY(instnorm1d_relu(h, a, n, S, cin, s));
        in1 = a;
      }
      SAMPT_TRY(conv1d(blk[i][0], in1, n, S, b, ACT_NONE, nullptr, s));
      SAMPT_TRY(instnorm1d_relu(b, b, n, S, cout, s));
      const bool fused_skip = cin == cout && i < 7;  // same width: the skip is the GEMM's residual operand
      SAMPT_TRY(conv1d(blk[i][1], b, n, S, c, ACT_NONE, fused_skip ? h : nullptr, s));
      if (!fused_skip) SAMPT_TRY(add_chanpad(c, h, (long)R, cin, cout, i == 7 ? 1 : 0, s));   // last block: + final_relu
      float* t = h;
      h = c, c = t;
    }
    GemmP d;
    d.A = h, d.W = dense_w, d.bias = dense_b, d.C = delta;
    d.M = (int)R, d.N = 2, d.K = 1024, d.lda = 1024, d.ldw = 1024, d.ldc = 2;
    SAMPT_TRY(gemm_f32(d, s));
    SAMPT_TRY(pips2_apply_delta(delta, bak, (float)stride, S, n, it == iters - 1, coords, trajs_out, s));
  }
  return SAMPT_OK;
}

}  // namespace sampt
