// Internal C++ launcher API of the HIP library (one function per kernel family).
// All tensors are device pointers owned by the caller; every launcher is stream-ordered, allocation-free and
// returns SAMPT_OK or a negative SAMPT_ERR_* code.
#pragma once
#include "common.h"

namespace sampt {

// ---- elementwise.hip ------------------------------------------------------------------------
// uint8 CHW frames (T,3,H,W) -> f32 NHWC4 (T,H,W,4) = 2*(x/255)-1, 4th channel 0      (pips.py:446)
int rgb_u8chw_to_nhwc4(const void* src, int src_f32, float* dst, int T, int H, int W, hipStream_t s);
// per-(image, channel) mean / rstd over H*W of an NHWC f32 tensor (InstanceNorm2d, eps, biased variance)
// partials: workspace of at least instnorm_partial_floats(nimg, hw, C) doubles
size_t instnorm_partial_doubles(int nimg, long hw, int C);
// elementwise.hip: row gather / scatter by (frame, object) item numbers and a plain fill (SamPt's decode staging)
int move_rows(const void* src, void* dst, const int* idx, int rows, long row_bytes, int nm, int nf, int scatter, hipStream_t s);
int fill_f32(float* dst, long n, float v, hipStream_t s);
// conv_stem_x3.hip: the tracker encoder's 7 x 7 stride-2 stem over NHWC4 frames as 3-term split-fp16 products (+ InstanceNorm partial sums)
int conv_stem_tiles(int H, int W);
int conv_stem7x7_x3(const float* x, const float* w, const float* bias, float* y, int nimg, int H, int W, double* in_part, hipStream_t s);
int instnorm_finalize(const double* partials, int nimg, int nchunks, long hw, int C, float eps, float* mean_rstd, hipStream_t s);
int instnorm_stats(const float* x, int nimg, long hw, int C, float eps, double* partials, float* mean_rstd,
                   hipStream_t s);
// y = (x-mean)*rstd ; if relu1: y = max(y,0) ; if skip: y = max(y + skip, 0)
// y_hi / y_lo (optional, both or none): y additionally split into two fp16 planes (hi = fp16(y), lo = fp16(y - hi)), the
// operand format of conv_f16x3's pre-split path.  y == nullptr (planes given): the f32 result is not written at all
int instnorm_apply(const float* x, const float* mean_rstd, const float* skip, float* y, int nimg, long hw, int C,
                   int relu1, hipStream_t s, half_t* y_hi = nullptr, half_t* y_lo = nullptr);
// bilinear resize of an NHWC f32 tensor into channels [c_off, c_off+C) of a dstC-channel NHWC tensor
// dst_hi / dst_lo (optional, both or none): write the result as two fp16 planes (hi, lo = the split-fp16 convolution's
// operand format) INSTEAD of the f32 dst
int resize_bilinear_nhwc(const float* src, int n, int sh, int sw, int C, float* dst, int dh, int dw, int dstC,
                         int c_off, int align_corners, hipStream_t s, half_t* dst_hi = nullptr, half_t* dst_lo = nullptr);
int avgpool2x2_nhwc(const float* src, int n, int h, int w, int C, float* dst, hipStream_t s);
// LayerNorm over the last dim of rows. src_rows: optional gather (row index into x, or -1 -> output row = 0).
// out_f16: 1 = y is half; 2 = y is half "x3 rows" [M][2D] (common.h GemmP::x3; D % 32 == 0). act: applied after the affine transform (ACT_GELU for LayerNorm2d+GELU).
// column means of an fp16 matrix over M (optionally gathered) rows; calibration of the fp16 ViT mode's bias correction
int colmean_rows_f16(const half_t* A, long M, int K, int lda, const int* rowmap, float* out, hipStream_t s);
int layernorm_rows(const float* x, const float* w, const float* b, void* y, long M, int D, float eps,
                   const int* src_rows, int out_f16, int act, hipStream_t s);
// out[i] = a[i] + b[i % bmod] (f32).  n, bmod in elements.
int add_bcast(const float* a, const float* b, float* out, long n, long bmod, hipStream_t s);
int cast_f32_f16(const float* x, half_t* y, long n, hipStream_t s);
// f32 rows [M][K] -> x3 rows [M][2K] halves (common.h GemmP::x3: per 32-block hi(32) | lo(32), saturating split)
int split_rows_x3(const float* x, half_t* y, long M, int K, hipStream_t s);
// SAM preprocess + patch im2col: frames u8 HWC (B,H,W,3) -> A[B*g*g][3*P*P] with k = c*P*P + ky*P + kx,
// value = (x-mean[c])/std[c] inside the h x w image, 0 in the padded region (Sam.preprocess, App. A-2)
// chw: frames are (B,3,H,W) planar instead of (B,H,W,3).  mean/stdv are HOST pointers (3 floats each).
int sam_patchify(const uint8_t* frames, int chw, int B, int H, int W, int img, int P, const float* mean,
                 const float* stdv, void* A, int out_f16, hipStream_t s);

// single-channel bilinear resize (align_corners=False) of n maps: the target_hw resize of sam_pt.py:205-206
int resize_logits(const float* src, int n, int sh, int sw, float* dst, int dh, int dw, hipStream_t s);
// uint8 index mask = argmax over {background 0, object logits [M][npix]} (vos_eval/eval.py:304-355)
int index_masks(const float* logits, int M, long npix, uint8_t* out, hipStream_t s);

// ---- attention.hip --------------------------------------------------------------------------
// Decomposed relative-position terms of SAM's ViT attention (App. A-3):
//   relh[bh][q][kh] = <q_vec, rel_pos_h[qh - kh + S-1]>, relw likewise.  qkv: [B*S*S][3*D] (f32 or f16),
//   outputs f32 [B*heads][S*S][S].
int vit_rel_bias(const void* qkv, int qkv_f16, const float* rel_h, const float* rel_w, int B, int S, int heads,
                 int hd, float* relh, float* relw, hipStream_t s);
// scores[bh][q][k] (already scaled) += relh[bh][q][k/S] + relw[bh][q][k%S]; softmax over k, in place (f32).
int softmax_rel_rows(float* scores, const float* relh, const float* relw, long BH, int Nq, int S, hipStream_t s);
// Window padding of SAM's window_partition inside the attention kernels: the B launches of S x S tokens are frames x nwin
// windows, window w % nwin = (wy, wx) in row-major order over nwx windows per row; token (iy, ix) of it is PADDING when
// wy*S + iy >= gh or wx*S + ix >= gw.  Padded tokens enter attention as keys with qkv = bias (zero padding comes after norm1,
// App. A-3): the kernels fetch their K / V from ``bias_row`` (the qkv bias as one row in the qkv matrix's own format: 3D halves,
// or an x3 row of 6D halves) and never read the qkv rows of padded tokens.  bias_row == nullptr: no padding anywhere.
struct FlashPad {
  const half_t* bias_row = nullptr;
  int nwx = 0, nwin = 0, gh = 0, gw = 0;
  // vit_flash_attention_f16 only, optional: the block's rel-pos tables as fp16 MFMA operand images (pack.rel_pos_operand_images:
  // [2][ceil((2 S - 1) / 32)][hd / 16][64][8] halves) — the prologue then loads 16 bytes per lane and k-step instead of converting
  // the f32 tables in every workgroup.  Same halves, same results.
  const half_t* rel_ops = nullptr;
};
// Fused flash-style attention, f16 operands, fp32 softmax/accumulate.  qkv f16 [B*S*S][3*D]; out f16 [B*S*S][D];
// rel_h / rel_w: rel_pos tables f32 [2S-1][hd] (the decomposed bias is computed inside the kernel).
int vit_flash_attention_f16(const half_t* qkv, const float* rel_h, const float* rel_w, half_t* out, int B, int S,
                            int heads, int hd, hipStream_t s, FlashPad pad = FlashPad());

// attention_x3.hip: the same attention at fp32 grade (3-term split-fp16 products).  qkv: x3 rows [B*S*S][2*3D] halves (common.h
// GemmP::x3), out: x3 rows [B*S*S][2*D]
int vit_flash_attention_x3(const half_t* qkv, const float* rel_h, const float* rel_w, half_t* out, int B, int S, int heads,
                           int hd, hipStream_t s, FlashPad pad = FlashPad());

// ---- kmedoids.hip (query-point selection, sam_pt/utils/query_points.py:62-99; bit-identical to query_points.kmedoids_alternate)
// xy: device [n][2] f32 pixel coordinates (integers), n <= 2048.  rowsums: out[i] = sum_j dist(i, j) (fp64, numpy's pairwise order)
int kmedoids_rowsums(const float* xy, int n, double* out, hipStream_t s);
// medoids: device int [K], in = the initial medoids, out = the converged ones; iters_out (device int, optional) = iterations run
int kmedoids_alternate(const float* xy, int n, int K, int* medoids, int max_iter, int* iters_out, hipStream_t s);

// ---- corners.hip (Shi-Tomasi query points + mask erosion, sam_pt/utils/query_points.py:102-194; bit-identical to the numpy
// restatement in sam_pt_amd/query_points.py)
size_t qp_corners_workspace_bytes(int H, int W);
// out = cv2.erode(mask, ones((k, k))) for a {0, 1} byte mask [H][W] (k = 0: the default 3 x 3; k = 1: a copy); tmp: H*W bytes
int qp_erode(const uint8_t* mask, int H, int W, int k, uint8_t* tmp, uint8_t* out, hipStream_t s);
// image u8 [3][H][W] RGB, mask u8 {0, 1} [H][W] -> up to n_points corners (x, y) f32 in out_xy [n_points][2]; out_info: 16 int32
// (device): [12] = corners found, [10] = k of the erosion that was kept (-1: none), [9] = pixels of the eroded mask, [5..8] its
// bounding box (ymin, ymax, xmin, xmax), [0..4] the mask's bounding box and pixel count
int qp_corners(const uint8_t* image, const uint8_t* mask, int H, int W, int n_points, float quality, float* out_xy, int* out_info,
               void* ws, size_t ws_bytes, hipStream_t s);

// ---- pips.hip ---------------------------------------------------------------------------------
struct PyramidLevels {
  const float* base[4];   // level l: [nframes][H_l][W_l][C] f32 NHWC
  int H[4], W[4];
};
// ffeat[n][c] = bilinear_sample2d(fmap[frame], x/stride..)   (samp.py:6-80)
// frame_idx: optional per-point frame index into fmap [frames][H][W][C]
int pips_sample_feat(const float* fmap, int H, int W, int C, const int* frame_idx, const float* xy, int n, float* out,
                     hipStream_t s);
// fused correlation + 7x7 window sampler (pips.py:364-407): writes x[n][s][xoff + lvl*49 + i*7+j]
// times != nullptr: the level-0 workgroups also do pips_build_input's job (PIPS layout: xoff == 128, 519 <= ldx <= 580)
int pips_corr_sample(const PyramidLevels& pyr, const int* frame_idx, int S, int n, int C, const float* ffeats,
                     const float* coords, float* x, int ldx, int xoff, hipStream_t s, const float* times = nullptr);
// mixer input assembly: x[n][s][0:128]=ffeats, x[...][324:519] = sincos embedding of (flow, t) (misc.py:30-55)
// times: device [S] = torch.linspace(0, S, S) (pips.py:527)
int pips_build_input(const float* ffeats, const float* coords, const float* times, int S, int n, float* x, int ldx,
                     hipStream_t s);
int vos_index_masks(const float* logits, int M, int T, long hw, const int* qt, const uint8_t* gt, uint8_t* out,
                    hipStream_t s);
int pil_resample_u8(const uint8_t* src, uint8_t* dst, long outer, int in_len, int out_len, int inner, const int* coef,
                    const int* bounds, int ksize, hipStream_t s);
int vos_index_masks_resized(const float* logits, int M, int T, int h, int w, const int* qt, const uint8_t* gt, int oh,
                            int ow, uint8_t* out, hipStream_t s);
int fill_rows_bias(void* out, int f16, const int* rows, int nrows, const float* bias, int N, hipStream_t s);
// ---- PIPS++ (pips2.hip; pips_plus_plus.py:263-342, 436-546) — rows are (point, frame): row = pt*S + s
int pips2_init(const float* trajs0, const float* fmap, int H, int W, const int* frame_idx, float stride, int S, int n,
               int have_init, float* coords, float* bak, float* f1, float* f2, float* f4, hipStream_t s);
int pips2_templates(const float* fmap, int H, int W, const int* frame_idx, const float* coords, int S, int n, float* f2,
                    float* f4, hipStream_t s);
int pips2_build_input(const float* coords, const float* omega, int S, int n, float* x, int ldx, hipStream_t s);
int instnorm1d_relu(const float* x, float* y, int n, int S, int C, hipStream_t s);
int add_chanpad(float* out, const float* identity, long rows, int cin, int cout, int relu, hipStream_t s);
int pips2_apply_delta(const float* delta, const float* bak, float stride, int S, int n, int last, float* coords,
                      float* trajs, hipStream_t s);
// coords[s][pt] = coords0[pt] = xys[pt]/stride ; ffeats[pt][s] = feat_init[pt]     (pips.py:458-476)
int pips_init_state(const float* xys, const float* feat_init, float stride, int S, int n, float* coords,
                    float* coords0, float* ffeats, hipStream_t s);
// token-mixing PreNormResidual block of the MLP-Mixer, grid (sequence, 64-channel chunk) (pips.py:116,120-121); out of
// place: xo != x
int pips_token_mix(const float* x, float* xo, const float* lnw, const float* lnb, const float* w1, const float* b1,
                   const float* w2, const float* b2, int nseq, int S, int D, hipStream_t s);
// device-side bookkeeping of the chained PIPS windows (pips/tracker.py:42-153; kernels and layouts: pips.hip)
int pips_chain_init(const float* q, int n, int T, int* cur, float* traj, float* vis, hipStream_t s);
int pips_round_begin(const int* cur, const unsigned char* flip, const float* traj, int T, int n, int S, int* fidx, float* xys,
                     float* xy_feat, int* f0, float stride, hipStream_t s);
int pips_round_end(int* cur, const float* tr, const float* vi, int T, int n, int S, float thr0, float* traj, float* vis,
                   int* n_active, hipStream_t s);
// ---- pips_mixer.hip: the mixer's channel MLP + everything between two channel MLPs, two launches per block (file header)
extern int g_pips_mixer_fused, g_pips_mixer_wgs, g_pips_mixer_diag;
// hidden slices (8, 16 or 32) a channel-MLP launch over nseq sequences uses to reach g_pips_mixer_wgs workgroups
int pips_mix_slices(int nseq);
// part[slice][nseq*8][512] = fc2 over the slice's hidden units of gelu(fc1(LN(x)) + b1); x [nseq*8][512]
int pips_mix_mlp(const float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                 float* part, int nseq, int NS, hipStream_t s);
// x' = res + (sum of the NS slabs + bias) (NS = 0: x' = res); mode 0: out [nseq*8][512] = x' + token-mix(LN(x')) with the
// token-mixing weights (w1 [32][8], w2 [8][32]); mode 1: out [nseq][512] = mean over the 8 tokens of LN(x').  out != res
int pips_mix_reduce(const float* part, int NS, const float* bias, const float* res, int nseq, int mode, const float* lnw,
                    const float* lnb, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
                    hipStream_t s);
// ---- pips_mixer_x3.hip: the same block with the channel MLP as 3-term split-fp16 products (file header)
extern int g_pips_mixer_x3;
size_t pips_mix_xop_halves(int nseq);            // size of the operand-image buffer between the two kernels
// x' = res + (slab sum + bias); xout = x' + token-mix(LN1(x')) (f32, the next residual); xop = operand images of 2^6 LN2(xout)
int pips_mix_pre(const float* part, int NS, const float* bias, const float* res, int nseq, const float* ln1w, const float* ln1b,
                 const float* tw1, const float* tb1, const float* tw2, const float* tb2, const float* ln2w, const float* ln2b,
                 float* xout, half_t* xop, hipStream_t s);
// part[slice][nseq*8][512] from the operand images and the packed weight stream of the block (pack.pips_mixer_x3_stream, NS = 16 | 32)
int pips_mix_mlp_x3(const half_t* xop, const half_t* wstream, const float* b1, float* part, int nseq, int NS, hipStream_t s);
// mean over the S tokens of LN(x): out[n][D]
int pips_ln_mean(const float* x, const float* lnw, const float* lnb, float* out, int nseq, int S, int D, hipStream_t s);
// feature / coordinate update (pips.py:536-544): delta [n][S][130]; ffeats [n][S][128]; coords [S][n][2]
// up_wT: ffeat_updater weight transposed to [in][out]
int pips_update(const float* delta, const float* gn_w, const float* gn_b, const float* up_wT, const float* up_b,
                float* ffeats, float* coords, const float* coords0, int S, int n, hipStream_t s);
// vis[s][n] = sigmoid(<ffeats[n][s], w> + b) ; traj[s][n][2] = coords*stride
int pips_finalize(const float* ffeats, const float* vis_w, const float* vis_b, const float* coords, float stride,
                  int S, int n, float* traj, float* vis, hipStream_t s);

// ---- cotracker.hip (CoTracker v1 windows: SURVEY.md App. A-6; layouts in the file header) -------------------------------
int cot_prepare(const float* qxy, const int* qt, const int* frame_map, float stride, int n, int T, float* xy0, int* fidx_pt,
                float* traj_out, float* vis_out, hipStream_t s);
int cot_window_init(int ind, int S_local, int prev, int na, int S, const int* qt, const float* xy0, const int* frame_map,
                    const float* coords_prev, const float* vis_prev, const float* feat_init, float* coords, float* visin,
                    float* mask, int* fidx, float* ffeats, hipStream_t s);
int cot_pos_embed(const float* coords, const float* pos_x, const float* pos_y, int H, int W, int E, int na, float* pos,
                  hipStream_t s);
int cot_build_input(const float* ffeats, const float* coords, const float* visin, const float* mask, const float* pos,
                    const float* times, int S, int na, float* x, hipStream_t s);
// attention over token groups taken from packed qkv rows [rows][3*heads*hd]: token t of group b = row b*bs + t*ts
int cot_attention(const float* qkv, float* out, int nbatch, int L, int bs, int ts, int heads, int hd, hipStream_t s);
int cot_window_store(const float* ffeats, const float* vis_w, const float* vis_b, const float* coords, float stride, int S,
                     int na, int ind, int S_local, int n_total, float* coords_prev, float* vis_prev, float* traj_out,
                     float* vis_out, hipStream_t s);
// F.interpolate(bilinear, align_corners=False) of n single-channel planes (uint8 or f32) to f32
int resize_planes(const void* src, int src_u8, long n, int sh, int sw, float* dst, int dh, int dw, hipStream_t s);

// ---- sam_decoder.hip (every launcher takes the frame batch F; tensors are [F][...] contiguous) ---------------
// decoder tokens [F][Nt][256] (Nt = 5 + k + (box ? 2 : 1)): out tokens, then the sparse prompt tokens (App. A-4).
// pts [F][ld_pts][2] input-frame px, labels [F][ld_pts] i32, box [F][4] or null.
int sam_tokens(const float* out_tokens /*[n_out][256]*/, int n_out, const float* pts, const int* labels, int k, int ld_pts,
               const float* box, const float* gauss, const float* point_emb /*[4][256]*/, const float* not_a_point,
               float img_size, int F, const int* k_item /*[F] or null*/, int* ntok /*[F] out or null*/, float* tokens,
               hipStream_t s);
// small f32 attention, one workgroup per (query, head, frame): q [F][Nq][H*hd], k,v [F][Nk][H*hd]
// nk_item: optional per-frame key count (<= Nk) of a ragged batch
int attn_rowblock(const float* q, const float* k, const float* v, float* out, int F, int Nq, int Nk, int heads, int hd,
                  const int* nk_item, hipStream_t s);
// small f32 attention with few keys (Nk <= 64): one thread per (query, head)
// token -> image attention, 8 heads x 16 channels (q/k/v/out rows of 128 floats), any Nq / Nk: keys split over workgroups,
// partial softmax states merged by a second launch; ws: attn_t2i_workspace_floats(F, Nq, Nk) floats (0: none needed)
size_t attn_t2i_workspace_floats(int F, int Nq, int Nk);
int attn_t2i(const float* q, const float* k, const float* v, float* out, int F, int Nq, int Nk, float* ws, size_t ws_floats,
             hipStream_t s, int ldkv = 128 /* row stride of k and v in floats (slices of a fused projection) */);
int attn_fewkeys(const float* q, const float* k, const float* v, float* out, int F, int Nq, int Nk, int heads, int hd,
                 const int* nk_item, hipStream_t s, int ldq = 0 /* row stride of q in floats (0: heads*hd) */);
// low_res[f][p] = <hyper[f][0:C], up[f][p][0:C]>
int sam_mask_dot(const float* up, const float* hyper, int ld_hyper, const float* up2 /*or null*/, const float* hyper2,
                 int ld_hyper2, float* low_res, int F, int npix, int C, hipStream_t s);
// fused Sam.postprocess_masks: low (L x L) -> bilinear to (img x img) -> crop (in_h,in_w) -> bilinear to (oh,ow)
int sam_postprocess(const float* low, int L, int img, int in_h, int in_w, float* out, int oh, int ow, hipStream_t s);
// bbox state per frame: int[5] = {xmin, ymin, xmax, ymax, count} of logits > 0 (refinement box of sam_pt.py:809-820);
// bbox_partial: scratch of F * bbox_partial_ints(oh, ow) ints (deterministic two-stage reduction, no atomics)
size_t bbox_partial_ints(int oh, int ow);
int sam_postprocess_bbox(const float* low, int L, int img, int in_h, int in_w, float* out, int oh, int ow, int F,
                         int* bbox, int* bbox_partial, hipStream_t s);
int bbox_from_logits_state(const float* logits, int h, int w, int* bbox_state, int* bbox_partial, hipStream_t s);
// mask-input embedding (PromptEncoder.mask_downscaling, App. A-4) fused with "src = image_embedding + dense":
//   mask (4g x 4g) -> conv2x2s2(1->c1) LN2d GELU -> conv2x2s2(c1->c2) LN2d GELU -> conv1x1(c2->256) ; src = feat + dense
struct MaskEmbedW {
  const float *w0, *b0, *ln0w, *ln0b, *w1, *b1, *ln1w, *ln1b, *w2, *b2;  // torch layouts [out][in][kh][kw]
};
int sam_mask_embed_src(const float* mask, int g, int F, const MaskEmbedW& w, const float* feat, float* tmp0,
                       float* tmp1, float* src, hipStream_t s);
// refinement gating (sam_pt.py:809-811): active[f] &= count(bbox_cur[f]) >= 2 ; box_f[f] = bbox_cur[f]
int sam_refine_gate(int* active, const int* bbox_cur, float* box_f, int F, bool first, hipStream_t s);
// where active[f]: cur <- cand for logits (n_logits), low-res (n_low), iou and the bbox state
int sam_commit(const int* active, const float* cand_logits, float* cur_logits, long n_logits, const float* cand_low,
               float* cur_low, long n_low, const float* cand_iou, float* cur_iou, const int* cand_bbox, int* cur_bbox,
               int F, hipStream_t s);
// out[f] = (iou[f] >= thr) ? logits[f] : -inf ; score_out[f] = iou[f]   (sam_pt.py:830-837)
int sam_finalize_mask(const float* logits, const float* iou, float thr, float* out, float* score_out, long n, int F,
                      hipStream_t s);

}  // namespace sampt
