// Host-side "engines": fixed launch sequences over the kernels in ops.h for the three stages of the SAM-PT hot
// path.  They own no device memory: weights are caller-owned device buffers looked up by their upstream
// checkpoint key, activations live in a caller-provided workspace carved by a bump allocator.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "ops.h"

namespace sampt {

struct WeightMap {
  std::unordered_map<std::string, const void*> m;
  mutable std::string missing;
  const void* get(const std::string& k) const {
    auto it = m.find(k);
    if (it == m.end()) {
      if (missing.size() < 900) missing += k + " ";
      return nullptr;
    }
    return it->second;
  }
  const float* f(const std::string& k) const { return (const float*)get(k); }
  const half_t* h(const std::string& k) const { return (const half_t*)get(k); }
  const int* i(const std::string& k) const { return (const int*)get(k); }
  bool has(const std::string& k) const { return m.count(k) != 0; }
};

// bump allocator over the caller's workspace; with base == nullptr it only measures.
struct Arena {
  char* base;
  size_t cap, off = 0, peak = 0;
  Arena(void* b, size_t c) : base((char*)b), cap(c) {}
  void* get(size_t bytes) {
    size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    if (off > peak) peak = off;
    return base ? (void*)(base + a) : (void*)(uintptr_t)(a + 256);  // fake non-null address when measuring
  }
  float* f32(size_t n) { return (float*)get(n * 4); }
  half_t* f16(size_t n) { return (half_t*)get(n * 2); }
  bool ok() const { return base == nullptr || peak <= cap; }
  bool dry() const { return base == nullptr; }
};

// -------------------------------------------------------------------------------------------------
struct ConvW {
  const float* w = nullptr;  // [Cout][KH*KW*Cin] (ci fastest), f32
  const float* b = nullptr;
  const half_t* w_hl = nullptr;  // optional: the same weights split into fp16 planes [2][Cout][K] (conv_f16x3.hip)
  int cin = 0, cout = 0, k = 0, stride = 1, pad = 0;
};

struct PipsEngine {
  int S = 8, stride = 4;
  int frames_f32 = 0;   // fnet input: 0 = uint8 frames, 1 = float frames in [0, 255] (PIPS++ with image_size)
  ConvW stem, conv2, conv3;
  ConvW blk[4][2][3];  // [layer][block][conv1, conv2, downsample]
  bool has_down[4][2] = {};
  // mixer
  const float *in_w, *in_b, *head_w, *head_b, *oln_w, *oln_b;
  struct MixBlk {
    const float *ln1w, *ln1b, *tw1, *tb1, *tw2, *tb2, *ln2w, *ln2b, *cw1, *cb1, *cw2, *cb2;
    const half_t* x3s[2];      // optional: the channel MLP's weights as split-fp16 operand streams for 16 / 32 hidden slices
  } mix[12];
  const float *gn_w, *gn_b, *up_wT, *up_b, *vis_w, *vis_b, *times;
  std::string error;
  int window_launches = 0;   // kernel launches of the last update() call (one window of every chain = the body of a round)

  int init(const WeightMap& w);
  int init_fnet(const WeightMap& w);   // encoder only (shared with PIPS++: same BasicEncoder, other stride)
  // frames: uint8 (nf,3,H,W).  out[l]: level-l feature maps [nf][H_l][W_l][128] f32 (H_0 = H/stride).
  int fnet(const uint8_t* frames, int nf, int H, int W, float* const out[4], Arena& ws, hipStream_t s);
  // one PIPS window per point: frame_idx (device, [n][S] ints into the pyramid), xys (device [n][2], px at frame 0),
  // feat_init (device [n][128]).  traj_out [S][n][2] px, vis_out [S][n] = sigmoid(logit).
  int update(const PyramidLevels& pyr, const int* frame_idx, int n, const float* xys, const float* feat_init, int iters,
             float* traj_out, float* vis_out, Arena& ws, hipStream_t s);
  // All chained windows of n point chains over a T-frame pyramid (pips/tracker.py:42-153), bookkeeping on the device.
  // q (device [n][3]): query (t, x, y) of each chain in ITS OWN time axis; flip (device [n] bytes): chain d-axis frame d
  // reads pyramid frame T-1-d.  q_host / flip_host: the same values on the host (which pyramid chunks a round may touch).
  // chunk_*: events recorded (on another stream) when pyramid frames [chunk_lo, chunk_hi) became valid; every round waits
  // for the chunks it can reach.  Rounds are enqueued one ahead of the device: the host learns through `flag` (pinned
  // host int32[2]) + `flag_ev[2]` that no chain is left, so this call synchronises with the stream and spends at most one
  // idle round.  traj [T][n][2] px, vis [T][n] sigmoid (device); *rounds = window rounds run.
  int track(const PyramidLevels& pyr, int T, int n, const float* q, const unsigned char* flip, const float* q_host,
            const unsigned char* flip_host, float thr0, int iters, void* const* chunk_ev, const int* chunk_lo,
            const int* chunk_hi, int nchunks, int* flag, hipEvent_t flag_ev[2], float* traj, float* vis, Arena& ws,
            hipStream_t s, int* rounds);
};

// PIPS++ (pips_plus_plus.py): the PIPS encoder at stride 8 + a 1-D ResNet over time instead of the MLP-Mixer; one call
// refines a whole chunk of S <= max_sequence_length frames for n points.
struct Pips2Engine {
  PipsEngine enc;                      // fnet only
  int stride = 8;
  struct Conv1 {
    const float *w, *b;                // [Cout][3*Cin] (tap-major, ci fastest), [Cout]
    int cin, cout;
  };
  Conv1 first, blk[8][2];
  const float *dense_w, *dense_b, *omega;
  std::string error;

  int init(const WeightMap& w, int stride);
  // frame_idx (device int [n][S]): pyramid frame of chunk frame s for point pt; trajs0 (device [S][n][2] px): initial
  // trajectory (zero velocity or the previous chunk's); feats[3] (device [n][S][128] each): templates, read as the
  // feat_init of the previous chunk when have_init != 0, always written back.  trajs_out [S][n][2] px.
  int update(const PyramidLevels& pyr, const int* frame_idx, int n, int S, const float* trajs0, int have_init,
             float* const feats[3], int iters, float* trajs_out, Arena& ws, hipStream_t s);
};

// CoTracker v1: the PIPS encoder at stride 4 + the UpdateFormer (6 time + 6 space attention blocks,
// hidden 384, 8 heads) refining sliding windows of 8 frames with step 4.
struct CotEngine {
  PipsEngine enc;  // fnet only
  int S = 8, stride = 4, hidden = 384, heads = 8, depth = 6;
  struct Blk {
    const float *qkv_w, *qkv_b, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  } tb[6], sb[6];
  const float *in_w, *in_b, *head_w, *head_b, *gn_w, *gn_b, *up_wT, *up_b, *vis_w, *vis_b;
  const float *times;             // [S][456] 1-D sin/cos time embedding (host-built: float64 -> f32 like upstream)
  const float *ln_one, *ln_zero;  // affine-free LayerNorm as weight 1 / bias 0
  std::string error;

  int init(const WeightMap& w);
  // One temporal direction of CoTracker.forward over T >= S model frames.  frame_map (device int [T]): pyramid frame of
  // model frame t (identity, reversed for the time-flipped pass, clamped for clips shorter than S).  Points are SORTED by
  // query frame: qt_host / qt_dev int [n] (the same values on the host — window membership is host control flow — and on
  // the device), qxy (device [n][2], px of the model's frame size).  pos_x [W0][228] / pos_y [H0][228]: the two 1-D tables
  // of the 2-D sin/cos position grid.  traj_out [T][n][2] px (0 where never written), vis_out [T][n] = sigmoid(logit)
  // (0.5 where never written).
  int track(const PyramidLevels& pyr, int T, const int* frame_map, int n, const int* qt_host, const int* qt_dev,
            const float* qxy, const float* pos_x, const float* pos_y, int iters, float* traj_out, float* vis_out, Arena& ws,
            hipStream_t s);
};

// -------------------------------------------------------------------------------------------------
struct VitConfig {
  int D = 768, depth = 12, heads = 12, grid = 64, window = 14, patch = 16, out_chans = 256, mlp_ratio = 4;
  int img = 1024;
  int global_mask = 0;  // bit i set -> block i uses global attention (depth <= 32)
  int f16 = 1;          // 1: fp16 MFMA GEMMs + flash attention ; 0: exact fp32 everywhere
  float mean[3] = {123.675f, 116.28f, 103.53f}, stdv[3] = {58.395f, 57.12f, 57.375f};
};

struct VitEngine {
  VitConfig c;
  struct Blk {
    const float *ln1w, *ln1b, *ln2w, *ln2b, *qkv_b, *proj_b, *b1, *b2, *rel_h, *rel_w;
    const half_t* rel_ops = nullptr;   // optional: rel_h / rel_w as fp16 MFMA operand images (pack.rel_pos_operand_images)
    const void *qkv_w, *proj_w, *w1, *w2;  // f16 or f32 depending on c.f16
    const half_t* qkv_b16 = nullptr;       // 16-bit modes: the qkv bias as one row of the qkv matrix's format (fp16 / x3 row)
  };
  std::vector<Blk> blk;
  const void *patch_w, *neck0_w, *neck2_w;
  const half_t *patch_hl = nullptr, *neck0_hl = nullptr, *neck2_hl = nullptr;   // split-fp16 planes (fast mode)
  const float *patch_b, *pos, *neck1w, *neck1b, *neck3w, *neck3b;
  const int* win_rows;  // [Bmax * nwin * window^2] -> token row or -1
  const int* win_inv;   // [Bmax * grid^2] token row -> window-order row
  const int* win_pad;   // [Bmax * npad] window-order rows that are zero padding
  const int* win_inv_live[8] = {};   // the same two maps for the compact live grids of k+1 window rows x grid columns
  const int* win_pad_live[8] = {};
  int win_rows_batches = 0;
  std::string error;
  // Optional in-situ timing of the fp16 GEMM launches (bench.py's roofline): HIP events on the launching stream around
  // every gemm_f16 call of encode() while `profiling` is set; collected (and destroyed) by profile_end().
  struct GemmEv {
    hipEvent_t a, b;
    double flop;
  };
  mutable std::vector<GemmEv> prof;
  bool profiling = false;
  // persistent workgroups per XCD of the fp16 GEMMs (0 = one per CU).  A 512-thread, 128-KiB GEMM workgroup owns its CU, so
  // a caller that runs latency-bound kernels on another stream beside the encoder (the point tracker) asks for fewer.
  int gemm_wgs = 0;
  // per launch kind (0 qkv, 1 proj, 2 fc1, 3 fc2): overrides gemm_wgs when > 0 (tile counts of the four shapes quantise
  // differently on a given number of workgroups: sampt_vit_set_gemm_workgroups_kind)
  int gemm_wgs_kind[4] = {0, 0, 0, 0};
  // Calibration of the fp16 mode's static bias correction (sampt_vit_calibrate): while set, every block GEMM of encode() also
  // writes the column means of its A operand (over the real tokens) to calib[(block * 4 + kind) * calib_ld + k].
  float* calib = nullptr;
  int calib_ld = 0;
  mutable int cur_blk = 0;
  int profile_end(double* flop, double* ms, int* launches);

  int init(const WeightMap& w, const VitConfig& cfg, int win_rows_batches);
  // frames: uint8 (B,3,H,W) if chw else (B,H,W,3); features out: [B][grid*grid][out_chans] f32 (NHWC);
  // interm_out (optional): [B][grid*grid][D] f32 = token stream after the first global-attention block (HQ-SAM's
  // interm_embeddings[0])
  // dead_mode 0: full computation.  1: B == 1, computes `dead_cache` [(grid - live_rows) * grid][D] f32 (the residual
  // stream of the frame-independent padding tokens at the input of the first global block) and returns.  2: blocks before
  // the first global one run on the live rows only, the dead rows come from `dead_cache` (see encode()).
  int encode(const uint8_t* frames, int chw, int B, int H, int W, float* features, float* interm_out, Arena& ws,
             hipStream_t s, float* dead_cache = nullptr, int dead_mode = 0);
  int live_rows(int H, int W) const;   // token rows that depend on the frame before the first global block (grid = all)
};

// -------------------------------------------------------------------------------------------------
struct DecConfig {
  int grid = 64, C = 256, heads = 8, depth = 2, mlp = 2048, img = 1024;
  int vit_dim = 0;  // > 0: HQ-SAM decoder (MaskDecoderHQ) fed by a ViT of that width
};

struct DecEngine {
  DecConfig c;
  int max_frames = 1;  // frame-batch capacity of the pixel-shuffle row maps
  struct Attn {
    const float *qw, *qb, *kw, *kb, *vw, *vb, *ow, *ob;
    int inner;
  };
  struct Layer {
    Attn self, t2i, i2t;
    const float *n1w, *n1b, *n2w, *n2b, *n3w, *n3b, *n4w, *n4b, *m1w, *m1b, *m2w, *m2b;
  } layer[4];
  Attn fin;
  const float *nfw, *nfb;
  const float *out_tokens;                       // [5][256] = iou_token, mask_tokens
  const float *gauss, *point_emb, *not_a_point, *no_mask, *dense_pe;
  MaskEmbedW me;
  const float *up0_w, *up0_b, *upln_w, *upln_b, *up1_w, *up1_b;  // ConvT weights packed [(dy,dx)][cout][cin]
  const half_t *up0_hl = nullptr, *up1_hl = nullptr;             // their split-fp16 planes [2][4*cout][cin] (optional)
  // Fused image-side projections (pack.py): per layer W = [t2i.k_proj; t2i.v_proj; i2t.q_proj] (384 x C), bias likewise,
  // pe = dense_pe @ [Wk; 0; Wq]^T (P x 384) — (keys + pe) W^T = keys W^T + pe W^T, so the positional term is a constant
  // added in the epilogue and the image tokens are read ONCE per layer; final attention: [k_proj; v_proj] (256 x C).
  struct FusedProj {
    const float *w = nullptr, *b = nullptr, *pe = nullptr;
    const half_t* hl = nullptr;
    int n = 0;
  } kvq[4], fin_kv;
  const int *up0_map, *up1_map;                  // pixel-shuffle row maps [4][max_frames*g*g], [4][max_frames*4*g*g]
  const float *hyp_w[3], *hyp_b[3], *iou_w[3], *iou_b[3];         // hypernetwork MLP 0 and IoU head
  const float *hypx_w[3][3], *hypx_b[3][3];                       // hypernetwork MLPs 1..3 (multimask_output=True)
  bool multimask = false;  // next decode(): SAM's multimask_output=True — 3 masks / IoUs of mask tokens 1..3 (F == 1, SAM only)
  // HQ-SAM extras (c.vit_dim > 0): out_tokens has a 6th row (hf_token); ConvT weights packed like up0/up1
  struct HqW {
    const float *mlp_w[3], *mlp_b[3];                             // hf_mlp
    const float *cv0_w, *cv0_b, *cvln_w, *cvln_b, *cv1_w, *cv1_b;  // compress_vit_feat  (vit_dim -> C -> C/8)
    const float *ee0_w, *ee0_b, *eeln_w, *eeln_b, *ee1_w, *ee1_b;  // embedding_encoder  (C -> C/4 -> C/8)
    const float *mf0_w, *mf0_b, *mfln_w, *mfln_b, *mf1_w, *mf1_b;  // embedding_maskfeature: conv3x3 [Cout][9*Cin]
    const half_t *mf0_hl = nullptr, *mf1_hl = nullptr;             // their split-fp16 planes (optional)
    const half_t *cv0_hl = nullptr, *cv1_hl = nullptr, *ee0_hl = nullptr, *ee1_hl = nullptr;
  } hq;
  // optional (opt-in, pack.pack_decoder with SAMPT_DEC_F16X3=1): split-fp16 planes [2][N][K] of the attention projection
  // weights, keyed by the f32 weight pointer; projections over the image tokens (M = F*g*g rows) then run on the fp16
  // matrix pipe at fp32 grade (conv_f16x3.hip as a 1x1 convolution over an [1][M][1][K] image)
  std::unordered_map<const float*, const half_t*> w_hl;
  std::string error;

  bool is_hq() const { return c.vit_dim > 0; }
  int n_out() const { return is_hq() ? 6 : 5; }
  int init(const WeightMap& w, const DecConfig& cfg);
  // HQ-SAM per-frame features [F][16*g*g][C/8] = embedding_encoder(features) + compress_vit_feat(interm)
  // (features [F][g*g][C], interm [F][g*g][vit_dim]); computed once per frame, shared by all decode passes.
  int hq_features(int F, const float* features, const float* interm, float* out, Arena& ws, hipStream_t s);
  // One predict_torch pass (multimask_output=False, return_logits=True) for F frames with the same prompt shape.
  // features [F][g*g][256]; pts [F][ld_pts][2], labels [F][ld_pts] (first k used); box [F][4] or null;
  // mask_in [F][4g][4g] or null.  logits_out [F][oh][ow], iou_out [F], low_out [F][4g][4g];
  // bbox_out: optional int [F][5] state of logits > 0.  hq_feat: hq_features() output (HQ-SAM) or null (SAM).
  // k_item: optional device int [F]: item f only uses its first min(k, k_item[f]) points (ragged batch: its tokens are
  // packed in front, padding rows are masked out of every attention).
  int decode(int F, const float* features, const float* hq_feat, const float* pts, const int* labels, int k,
             const int* k_item, int ld_pts, const float* box, const float* mask_in, int in_h, int in_w, int oh, int ow, float* logits_out, float* iou_out, float* low_out,
             int* bbox_out, Arena& ws, hipStream_t s);
  // The whole per-(frame, object) chain of SamPt.predict_mask (sam_pt.py:760-837) for F frames, no host sync:
  // [positives-only pass when n_pos_first >= 0] -> all-points pass (+ mask) -> R box-refinement passes gated per frame
  // on device -> IoU-threshold rejection.  final_logits [F][oh][ow], score_out [F].
  // k_item / npos_item: optional per-item point counts (all points / leading positives) of a ragged batch.
  int track_decode(int F, const float* features, const float* hq_feat, const float* pts, const int* labels, int k,
                   const int* k_item, const int* npos_item, int ld_pts, int n_pos_first, int R, float iou_thr, int in_h, int in_w, int oh, int ow, float* final_logits, float* score_out,
                   Arena& ws, hipStream_t s);
};

}  // namespace sampt
