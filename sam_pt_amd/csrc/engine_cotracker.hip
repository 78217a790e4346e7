// CoTracker engine: the whole sliding-window pass of CoTracker.forward for one temporal direction
// as ONE fixed launch sequence.  Which points are active in which window depends only on the query frames, so — unlike
// PIPS' visibility-driven chaining — there is no host decision between windows: the carry-over of coordinates and
// visibility logits from window to window stays on the device and the host never synchronises.
//
// Reference call site: sam_pt/point_tracker/cotracker/tracker.py:104 (`self.model(rgbs, queries, iters=6)`) and :159-161
// (time-flipped pass); the model itself is third-party (co-tracker @ 4f297a9).
#include "engine.h"

namespace sampt {

int CotEngine::init(const WeightMap& w) {
  enc.stride = stride;
  enc.frames_f32 = 1;  // the adapter feeds the model a float video (bilinear resize to interp_shape)
  int rc = enc.init_fnet(w);
  if (rc != SAMPT_OK) {
    error = enc.error;
    return rc;
  }
  const std::string u = "updateformer.";
  in_w = w.f(u + "input_transform.weight"), in_b = w.f(u + "input_transform.bias");
  head_w = w.f(u + "flow_head.weight"), head_b = w.f(u + "flow_head.bias");
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < depth; ++i) {
      const std::string p = u + (k == 0 ? "time_blocks." : "space_blocks.") + std::to_string(i);
      Blk& b = k == 0 ? tb[i] : sb[i];
      b.qkv_w = w.f(p + ".attn.qkv.weight"), b.qkv_b = w.f(p + ".attn.qkv.bias");
      b.proj_w = w.f(p + ".attn.proj.weight"), b.proj_b = w.f(p + ".attn.proj.bias");
      b.fc1_w = w.f(p + ".mlp.fc1.weight"), b.fc1_b = w.f(p + ".mlp.fc1.bias");
      b.fc2_w = w.f(p + ".mlp.fc2.weight"), b.fc2_b = w.f(p + ".mlp.fc2.bias");
    }
  gn_w = w.f("norm.weight"), gn_b = w.f("norm.bias");
  up_wT = w.f("ffeat_updater.0.weight_t"), up_b = w.f("ffeat_updater.0.bias");
  vis_w = w.f("vis_predictor.0.weight"), vis_b = w.f("vis_predictor.0.bias");
  times = w.f("__times_embed"), ln_one = w.f("__ln_ones"), ln_zero = w.f("__ln_zeros");
  if (!w.missing.empty()) {
    error = "CotEngine: missing weights: " + w.missing;
    return SAMPT_ERR_ARG;
  }
  return SAMPT_OK;
}

namespace {
struct Lin {
  hipStream_t s;
  float* skws;
  size_t skn;
  int operator()(const float* A, int M, int K, const float* W, const float* b, float* C, int N, int act = ACT_NONE,
                 const float* res = nullptr) const {
    GemmP p;
    p.A = A, p.W = W, p.bias = b, p.C = C, p.res = res;
    p.splitk_ws = skws, p.splitk_ws_floats = skn;
    p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldc = N, p.ldr = N, p.act = act;
    return gemm_f32(p, s);
  }
};
}  // namespace

int CotEngine::track(const PyramidLevels& pyr, int T, const int* frame_map, int n, const int* qt_host, const int* qt_dev,
                     const float* qxy, const float* pos_x, const float* pos_y, int iters, float* traj_out, float* vis_out,
                     Arena& ws, hipStream_t s) {
  const bool dry = ws.dry();
  const int D = hidden, E = 456, R = n * S, hd = hidden / heads;
  float* coords = ws.f32((size_t)S * n * 2);
  float* coords_prev = ws.f32((size_t)S * n * 2);
  float* visin = ws.f32((size_t)S * n);
  float* vis_prev = ws.f32((size_t)S * n);
  float* mask = ws.f32((size_t)S * n);
  int* fidx = (int*)ws.get((size_t)n * S * sizeof(int));
  float* xy0 = ws.f32((size_t)n * 2);
  int* fidx_pt = (int*)ws.get((size_t)n * sizeof(int));
  float* feat_init = ws.f32((size_t)n * 128);
  float* ffeats = ws.f32((size_t)R * 128);
  float* x = ws.f32((size_t)R * E);
  float* pos = ws.f32((size_t)n * E);
  float* h = ws.f32((size_t)R * D);
  float* lnb = ws.f32((size_t)R * D);
  float* qkv = ws.f32((size_t)R * 3 * D);
  float* att = ws.f32((size_t)R * D);
  float* hid = ws.f32((size_t)R * 4 * D);
  float* delta = ws.f32((size_t)R * 130);
  const size_t skn = (size_t)8 * R * 4 * D;
  float* skws = ws.f32(skn);
  if (!ws.ok()) return SAMPT_ERR_WORKSPACE;
  if (dry) return SAMPT_OK;
  if (T < S || n <= 0) return SAMPT_ERR_ARG;
  for (int i = 1; i < n; ++i)
    if (qt_host[i] < qt_host[i - 1]) return SAMPT_ERR_ARG;          // points sorted by query frame (CoTracker.forward)
  if (qt_host[0] < 0 || qt_host[n - 1] >= T) return SAMPT_ERR_ARG;
  const Lin lin{s, skws, skn};
  SAMPT_TRY(cot_prepare(qxy, qt_dev, frame_map, (float)stride, n, T, xy0, fidx_pt, traj_out, vis_out, s));
  // feature of every point at its own query frame and position (bilinear_sample2d on the stride-4 map)
  SAMPT_TRY(pips_sample_feat(pyr.base[0], pyr.H[0], pyr.W[0], 128, fidx_pt, xy0, n, feat_init, s));
  int prev = 0;
  for (int ind = 0; ind < T - S / 2; ind += S / 2) {
    const int S_local = T - ind < S ? T - ind : S;
    int na = 0;
    while (na < n && qt_host[na] < ind + S) ++na;
    if (na == 0) continue;
    const int Ra = na * S;
    SAMPT_TRY(cot_window_init(ind, S_local, prev, na, S, qt_dev, xy0, frame_map, coords_prev, vis_prev, feat_init, coords,
                              visin, mask, fidx, ffeats, s));
    SAMPT_TRY(cot_pos_embed(coords, pos_x, pos_y, pyr.H[0], pyr.W[0], E, na, pos, s));
    for (int it = 0; it < iters; ++it) {
      SAMPT_TRY(pips_corr_sample(pyr, fidx, S, na, 128, ffeats, coords, x, E, 130, s));
      SAMPT_TRY(cot_build_input(ffeats, coords, visin, mask, pos, times, S, na, x, s));
      SAMPT_TRY(lin(x, Ra, E, in_w, in_b, h, D));
      for (int i = 0; i < depth; ++i)
        for (int k = 0; k < 2; ++k) {   // time block (tokens of one point), then space block (tokens of one frame)
          const Blk& b = k == 0 ? tb[i] : sb[i];
          SAMPT_TRY(layernorm_rows(h, ln_one, ln_zero, lnb, Ra, D, 1e-6f, nullptr, 0, ACT_NONE, s));
          SAMPT_TRY(lin(lnb, Ra, D, b.qkv_w, b.qkv_b, qkv, 3 * D));
          if (k == 0) SAMPT_TRY(cot_attention(qkv, att, na, S, S, 1, heads, hd, s));
          else SAMPT_TRY(cot_attention(qkv, att, S, na, 1, S, heads, hd, s));
          SAMPT_TRY(lin(att, Ra, D, b.proj_w, b.proj_b, h, D, ACT_NONE, h));
          SAMPT_TRY(layernorm_rows(h, ln_one, ln_zero, lnb, Ra, D, 1e-6f, nullptr, 0, ACT_NONE, s));
          SAMPT_TRY(lin(lnb, Ra, D, b.fc1_w, b.fc1_b, hid, 4 * D, ACT_GELU_TANH));
          SAMPT_TRY(lin(hid, Ra, 4 * D, b.fc2_w, b.fc2_b, h, D, ACT_NONE, h));
        }
      SAMPT_TRY(lin(h, Ra, D, head_w, head_b, delta, 130));
      SAMPT_TRY(pips_update(delta, gn_w, gn_b, up_wT, up_b, ffeats, coords, nullptr, S, na, s));   // no frame-0 lock
    }
    SAMPT_TRY(cot_window_store(ffeats, vis_w, vis_b, coords, (float)stride, S, na, ind, S_local, n, coords_prev, vis_prev,
                               traj_out, vis_out, s));
    prev = na;
  }
  return SAMPT_OK;
}

}  // namespace sampt
