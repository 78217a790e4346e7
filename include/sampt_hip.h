/* sampt_hip.h — C ABI of libsampt_hip.so, the MI355X (gfx950) HIP implementation of the SAM-PT hot path.
 *
 * The reference (SysCV/sam-pt) has no native code and no FFI: its "plugin boundary" is duck typing through Hydra
 * `_target_`s.  This header is therefore the binding a maintainer of the reference would add
 * underneath the two Python seams — see INTEGRATION.md for the ctypes stub:
 *
 *   seam 1  sam_pt.point_tracker.PointTracker.forward(rgbs, query_points)        sam_pt/point_tracker/tracker.py:26-51
 *           as implemented by PipsPointTracker                                   sam_pt/point_tracker/pips/tracker.py:42-201
 *   seam 2  SamPredictor.set_image / predict_torch as driven by SamPt            sam_pt/modeling/sam_pt.py:771, 783-828, 849
 *
 * Conventions: every function returns 0 (SAMPT_OK) or a negative error code and never throws; all pointers named
 * `*_dev` / documented "device" are HIP device pointers owned by the caller; kernels never allocate; work is
 * enqueued on the given hipStream_t (passed as void*) and no function synchronises unless documented.  Activation
 * scratch comes from a caller-provided workspace: call the *_workspace_bytes query first.
 */
#ifndef SAMPT_HIP_H
#define SAMPT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAMPT_OK 0
#define SAMPT_ERR_ARG (-1)
#define SAMPT_ERR_HIP (-2)
#define SAMPT_ERR_UNSUPPORTED (-3)
#define SAMPT_ERR_WORKSPACE (-4)

typedef void* sampt_stream_t; /* hipStream_t */
typedef struct sampt_pips* sampt_pips_t;
typedef struct sampt_pips2* sampt_pips2_t;
typedef struct sampt_cotracker* sampt_cotracker_t;
typedef struct sampt_vit* sampt_vit_t;
typedef struct sampt_dec* sampt_dec_t;

int sampt_version(void);
/* Last error message of the calling thread's most recent failing call ("" if none). */
const char* sampt_last_error(void);

/* ---------------------------------------------------------------------------------------------------------
 * seam 1 — PIPS point tracker.  Weights: `n` (name, device pointer) pairs keyed by the upstream checkpoint keys
 * (model-*.pth['model_state_dict'], module tree pips.py:191-287, 290-317, 410-437) after the host-side repack
 * documented in sam_pt_amd/pack.py (conv weights -> [Cout][KH][KW][Cin], stem Cin padded to 4, mixer input
 * weight K padded 519 -> 520, "ffeat_updater.0.weight_t", "__times").
 * --------------------------------------------------------------------------------------------------------- */
int sampt_pips_create(const char* const* names, const void* const* ptrs, int n, int stride, int S, sampt_pips_t* out);
void sampt_pips_destroy(sampt_pips_t h);

/* Pips.fnet (BasicEncoder, pips.py:254-287) + CorrBlock pyramid (pips.py:355-361) for `nf` frames.
 * frames_dev: uint8 (nf,3,H,W) as handed to PointTracker.forward.  pyr_dev[l]: float32 [nf][H_l][W_l][128] (NHWC),
 * H_0 = H/stride, H_{l+1} = H_l/2.  Replaces `self.fnet(rgbs_)` at pips.py:453-455 and CorrBlock.__init__. */
int sampt_pips_fnet_workspace_bytes(sampt_pips_t h, int nf, int H, int W, size_t* bytes);
int sampt_pips_fnet_f32(sampt_pips_t h, const uint8_t* frames_dev, int nf, int H, int W, float* const pyr_dev[4],
                        void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);

/* Initial point features: bilinear_sample2d(fmaps[:,0], xy/stride) (pips.py:469-475, utils/samp.py:6-80).
 * fmap_dev: level-0 maps [frames][H0][W0][128]; frame_idx_dev: int32 [n] frame of each point (NULL: frame 0);
 * xy_dev: [n][2] in feature-map pixels; out_dev: [n][128]. */
int sampt_pips_sample_feat_f32(const float* fmap_dev, int H0, int W0, const int32_t* frame_idx_dev, const float* xy_dev,
                               int n, float* out_dev, sampt_stream_t stream);

/* All chained 8-frame windows of PipsPointTracker._forward (sam_pt/point_tracker/pips/tracker.py:42-153) for n point CHAINS
 * (one point in one temporal direction) over a T-frame pyramid, with the per-window bookkeeping — window frames, write-back
 * of frames 1..7, visibility-threshold linking (tracker.py:111-148) — on the device.  q_* [n][3] = (t, x, y) of each chain's
 * query in ITS OWN time axis (the same values on the device and on the host), flip_* [n] bytes: chain frame d reads pyramid
 * frame T-1-d (tracker.py:162-167).  vis_threshold = initial_next_frame_visibility_threshold (pips.yaml:5).  chunk_events
 * (hipEvent_t[nchunks], may be NULL with nchunks = 0): pyramid frames [chunk_lo[c], chunk_hi[c]) are valid once event c has
 * fired; each round waits only for the chunks it can reach.  traj_dev [T][n][2] px and vis_dev [T][n] (sigmoid; 0 where
 * never written) are complete when the call returns: the number of rounds is data dependent, so this entry point
 * SYNCHRONISES with `stream` (rounds are enqueued one ahead of the device; at most one idle round is spent). */
int sampt_pips_track_workspace_bytes(sampt_pips_t h, int n, size_t* bytes);
int sampt_pips_track_f32(sampt_pips_t h, const float* const pyr_dev[4], int H0, int W0, int T, int n, const float* q_dev,
                         const uint8_t* flip_dev, const float* q_host, const uint8_t* flip_host, float vis_threshold, int iters,
                         void* const* chunk_events, const int32_t* chunk_lo, const int32_t* chunk_hi, int nchunks,
                         float* traj_dev, float* vis_dev, void* ws, size_t ws_bytes, sampt_stream_t stream, int32_t* rounds);
/* One 8-frame window of Pips.forward's iterative update (pips.py:458-476, 507-568) + sigmoid (pips/tracker.py:102).
 * frame_idx_dev: int32 [n][S] — per POINT, the indices of its window frames in the pyramid (tail repeated as
 * tracker.py:73-78).  Points are independent in PIPS, so points anchored at different frames share one call;
 * xys_dev: [n][2] pixels at the window's first frame; feat_init_dev: [n][128].
 * traj_out_dev: [S][n][2] pixels (last iteration); vis_out_dev: [S][n] in (0,1). */
/* Kernel launches of one ROUND of the last sampt_pips_track_f32 call (one window of every chain: round begin, the window's
 * iterations as counted at their launch sites, round end); 0 before the first call. */
int sampt_pips_round_launches(sampt_pips_t h);
int sampt_pips_update_workspace_bytes(sampt_pips_t h, int n, size_t* bytes);
int sampt_pips_update_f32(sampt_pips_t h, const float* const pyr_dev[4], int H0, int W0, const int32_t* frame_idx_dev,
                          int n, const float* xys_dev, const float* feat_init_dev, int iters, float* traj_out_dev,
                          float* vis_out_dev, void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * seam 1b — PIPS++ point tracker (sam_pt/point_tracker/pips_plus_plus/{pips_plus_plus.py:420-546, tracker.py:25-134}).
 * Weights keyed by the checkpoint names after sam_pt_amd.pack.pack_pips2 (fnet convs as for PIPS; Conv1d weights
 * (Cout, Cin, 3) -> [Cout][3][Cin] with the first conv's 718 input channels zero-padded to 720; "__omega" [32]).
 * fnet_f32 = PipsPlusPlus.fnet on every frame + the 4-level average-pool pyramid of CorrBlock.__init__ (:366-379),
 * stride 8.  update_f32 = one PipsPlusPlus.forward after the encoder for n points over a chunk of S frames
 * (frame_idx_dev int32 [n][S] selects each point's pyramid frames, so sub-clips and time-reversed clips are index
 * maps): trajs0_dev [S][n][2] px is `trajs_e0`, feats_dev[3] = (feats1, feats2, feats4) [n][S][128] are read as
 * `feat_init` when have_feat_init != 0 and always written back (tracker.py:49-54), trajs_out_dev [S][n][2] px =
 * coord_predictions1[-1].
 * --------------------------------------------------------------------------------------------------------- */
int sampt_pips2_create(const char* const* names, const void* const* ptrs, int n, int stride, sampt_pips2_t* out);
void sampt_pips2_destroy(sampt_pips2_t h);
int sampt_pips2_fnet_workspace_bytes(sampt_pips2_t h, int nf, int H, int W, size_t* bytes);
/* frames_dev: (nf,3,H,W) uint8, or float32 in [0, 255] when frames_are_f32 != 0 (the bilinearly pre-resized video of
 * PipsPlusPlusPointTracker(image_size=...), tracker.py:69-76). */
int sampt_pips2_fnet_f32(sampt_pips2_t h, const void* frames_dev, int frames_are_f32, int nf, int H, int W,
                         float* const pyr_dev[4], void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
int sampt_pips2_update_workspace_bytes(sampt_pips2_t h, int n, int S, size_t* bytes);
int sampt_pips2_update_f32(sampt_pips2_t h, const float* const pyr_dev[4], int H0, int W0, const int32_t* frame_idx_dev,
                           int n, int S, const float* trajs0_dev, int have_feat_init, float* const feats_dev[3],
                           int iters, float* trajs_out_dev, void* workspace_dev, size_t workspace_bytes,
                           sampt_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * seam 1c — CoTracker = what sam_pt.point_tracker.cotracker.CoTrackerPointTracker.forward (cotracker/tracker.py:72-170)
 * asks of the third-party model (facebookresearch/co-tracker @ 4f297a9, built by build_cotracker from
 * cotracker_stride_4_wind_8.pth: CoTracker(stride=4, S=8, space_depth=6, time_depth=6)).
 * Weight names: the checkpoint's keys ("fnet.*" packed like PIPS' encoder, "updateformer.input_transform",
 * "updateformer.{time,space}_blocks.{i}.{attn.qkv,attn.proj,mlp.fc1,mlp.fc2}", "updateformer.flow_head", "norm",
 * "ffeat_updater.0" (+ ".weight_t"), "vis_predictor.0") plus "__times_embed" [8][456], "__ln_ones"/"__ln_zeros" [384]
 * (sam_pt_amd/pack.py::pack_cotracker).
 *   resize_frames_f32 : the adapter's F.interpolate(rgbs, interp_shape, mode="bilinear") (tracker.py:90-92) over `planes`
 *                       = T*3 single-channel planes, uint8 (src_u8 != 0) or float32 input, float32 output in [0, 255].
 *   fnet_f32          : CoTracker.fnet (the BasicEncoder of PIPS, stride 4) on every float frame + the 4-level average-pool
 *                       pyramid of CorrBlock; each frame once per clip (InstanceNorm is per-sample), both directions share it.
 *   track_f32         : one CoTracker.forward (one temporal direction) over T >= 8 model frames: sliding windows of 8 frames
 *                       with step 4, `iters` UpdateFormer iterations each, the carry-over between windows on the device, no
 *                       host synchronisation.  frame_map_dev int32 [T]: pyramid frame of model frame t (identity; T-1-t for
 *                       the time-flipped pass of tracker.py:154-161; clamped to the last real frame for clips shorter than 8,
 *                       tracker.py:17-21).  Points SORTED by query frame as CoTracker.forward sorts them: query_t_host /
 *                       query_t_dev int32 [n] (same values; window membership is host control flow), query_xy_dev [n][2] px
 *                       of the model frame size.  pos_x_dev [W0][228] / pos_y_dev [H0][228]: the 1-D tables of
 *                       get_2d_sincos_pos_embed(456, (H0, W0)).  traj_out_dev [T][n][2] px — exact zeros where no window
 *                       wrote (what the adapter's `== 0` back-fill keys on, tracker.py:166) — vis_out_dev [T][n] =
 *                       sigmoid(visibility logit), 0.5 where no window wrote.
 * --------------------------------------------------------------------------------------------------------- */
int sampt_cotracker_create(const char* const* names, const void* const* ptrs, int n, int stride, int S,
                           sampt_cotracker_t* out);
void sampt_cotracker_destroy(sampt_cotracker_t h);
int sampt_resize_frames_f32(const void* frames_dev, int src_u8, long planes, int H, int W, float* out_dev, int out_h,
                            int out_w, sampt_stream_t stream);
int sampt_cotracker_fnet_workspace_bytes(sampt_cotracker_t h, int nf, int H, int W, size_t* bytes);
int sampt_cotracker_fnet_f32(sampt_cotracker_t h, const float* frames_dev, int nf, int H, int W, float* const pyr_dev[4],
                             void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
int sampt_cotracker_track_workspace_bytes(sampt_cotracker_t h, int n, size_t* bytes);
int sampt_cotracker_track_f32(sampt_cotracker_t h, const float* const pyr_dev[4], int H0, int W0, int T,
                              const int32_t* frame_map_dev, int n, const int32_t* query_t_host,
                              const int32_t* query_t_dev, const float* query_xy_dev, const float* pos_x_dev,
                              const float* pos_y_dev, int iters, float* traj_out_dev, float* vis_out_dev,
                              void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * seam 2a — SAM image encoder = SamPredictor.set_image (Sam.preprocess + ImageEncoderViT, Appendix A-1..A-3).
 * cfg: see sampt_vit_config.  Weight names: upstream sam_vit_*.pth keys; GEMM weights as ".f16" copies when
 * cfg.f16 == 1, as ".x3" copies (x3 rows of w * 2^8, see sampt_gemm_ex) when cfg.f16 == 2;
 * "image_encoder.neck.2.weight_khwc"; "__win_rows" (window-partition row map for win_batches frames).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct sampt_vit_config {
  int embed_dim, depth, num_heads, grid, window, patch, out_chans, mlp_ratio, img_size;
  int global_mask; /* bit i: block i uses global attention (configs/model/sam/image_encoder/vit_*.yaml) */
  int f16;         /* 1: fp16 MFMA + flash attention (fp32 accumulate/softmax/LN/residual); 0: exact fp32 (f32 MFMA,
                      materialised scores); 2: "f16x3" — the fp16 graph with every product rebuilt from split-fp16 pieces
                      (hi.hi + hi.lo + lo.hi, fp32 accumulate): fp32-grade results on the fp16 matrix pipe */
  float pixel_mean[3], pixel_std[3];
} sampt_vit_config;

int sampt_vit_create(const sampt_vit_config* cfg, const char* const* names, const void* const* ptrs, int n,
                     int win_batches, sampt_vit_t* out);
void sampt_vit_destroy(sampt_vit_t h);
int sampt_vit_encode_workspace_bytes(sampt_vit_t h, int B, size_t* bytes);
/* Persistent workgroups per XCD (of 32 CUs) for the encoder's fp16 GEMMs from now on; 0 = one per CU (the default).  A
 * GEMM workgroup (512 threads, 128 KiB of LDS, 256 VGPRs) owns its CU: a caller that runs latency-bound kernels on another
 * stream beside the encoder (the point tracker's window rounds) leaves them a few CUs per XCD this way. */
int sampt_vit_set_gemm_workgroups(sampt_vit_t h, int per_xcd);
/* The same per launch kind — the qkv, proj, fc1 and fc2 GEMM of a block — each 0 .. 32, 0 = whatever
 * sampt_vit_set_gemm_workgroups says.  The four shapes have different tile counts (N / 256 = 15, 5, 20, 5 column tiles for
 * ViT-H), so the number of workgroups that wastes least of the last round of tiles differs per kind. */
int sampt_vit_set_gemm_workgroups_kind(sampt_vit_t h, int qkv, int proj, int fc1, int fc2);
/* Process-wide experiment knob of the 8-phase fp16 GEMM (csrc/gemm_f16_p8.hip): G > 1 phase groups — persistent workgroup w of an
 * XCD belongs to group w % G and starts (w % G) / G of a tile's K-loop late, so that the groups' epilogue bursts interleave with
 * the other groups' K-loops.  0 / 1 = off (the default). */
int sampt_gemm_set_stagger(int groups);
/* Process-wide A / B switch of the same kernel's stage schedule: 0 (default) = the LDS-DMA instructions of a phase are issued in
 * its read segment, two phases after the half tile's last read; 1 = behind the first MFMAs of its multiply segment, one phase
 * after the last read (the round-3 / round-4 schedule).  Results are bitwise identical. */
int sampt_gemm_set_schedule(int sched);
/* Process-wide switch of the persistent fp16 GEMM's launcher: 1 (default) = a launch uses the FEWEST workgroups per XCD that need the
 * same number of tile rounds as the allowed count (sampt_vit_set_gemm_workgroups, or 32) — same finishing time, whole CUs left to the
 * streams beside it; 0 = always the allowed count.  Results are bitwise identical. */
int sampt_gemm_set_trim(int on);
/* Process-wide knob of the thin f32 GEMM (csrc/gemm.hip gemm_thin_f32: the tracker mixers' token-side products): the launcher grows
 * the (16 * FM) x 16 tile only while at least n workgroups remain (default 256 = one per CU). */
int sampt_gemm_set_thin_min_wgs(int n);
/* Process-wide A / B switch: 1 (default) = 3 x 3 stride-1 split-fp16 convolutions over pre-split planes (the tracker encoder's) run on
 * the halo-tiled kernel of csrc/conv_halo_x3.hip; 0 = on the implicit-GEMM LDS-DMA kernel of rounds 3 - 5 (csrc/conv_f16x3.hip);
 * 2 = halo-tiled with 4-wave workgroups at every tile width (1 uses 8 waves from 96 output channels up); 3 = as 1, but the tracker
 * encoder's InstanceNorms sum their statistics in a pass of their own instead of in the convolution's epilogue. */
int sampt_conv_set_halo(int on);
/* Staging of SamPt's per-frame decode (sam_pt.py:690-757 _apply_sam_to_trajectories: one predict_mask per (frame, object) item,
 * item = frame * n_objects + object) for a batched chain that reads and writes fixed buffers.  idx: rows int32 item numbers (device).
 *   scatter = 0: dst[r] = src[idx[r] / n_objects]                                   (item r's frame embedding into the chain's input)
 *   scatter = 1: dst[(idx[r] % n_objects) * n_frames + idx[r] / n_objects] = src[r]  (masks into logits [n_objects][n_frames][H * W];
 *                scores with n_objects = 1: dst[idx[r]] = src[r])
 * row_bytes: a multiple of 16 with 16-byte aligned buffers, or 4.  sampt_fill_f32: dst[0 .. n) = value (the -inf of items without
 * a prompt, sam_pt.py:766-767). */
int sampt_move_rows(const void* src, void* dst, const int* idx, int rows, size_t row_bytes, int n_objects, int n_frames, int scatter,
                    sampt_stream_t stream);
int sampt_fill_f32(float* dst, size_t n, float value, sampt_stream_t stream);
/* The tracker encoder's stem (pips.py:200 BasicEncoder.conv1: Conv2d(3, 64, 7, stride 2, padding 3)) over normalised NHWC4 frames
 * x [n][H][W][4] (fourth channel 0) with w f32 [64][7][7][4], as 3-term split-fp16 MFMA products — fp32-grade.  y [n][OH][OW][64].
 * mean_rstd non-null: also the statistics of the InstanceNorm2d that follows ([n][64][2], as sampt_conv3x3_planes_instnorm_stats);
 * ws then holds n * ceil(OH / 16) * ceil(OW / 16) * 64 * 16 bytes. */
int sampt_conv_stem7x7(const float* x_nhwc4, const float* w, const float* bias, float* y, int n, int H, int W, float eps,
                       float* mean_rstd, void* ws, size_t ws_bytes, sampt_stream_t stream);
/* 3 x 3 stride-1 pad-1 split-fp16 convolution over pre-split planes (sampt_conv2d_nhwc dtype 4) that ALSO produces the statistics of
 * the InstanceNorm2d that follows it in the tracker's encoder (pips.py:191-287 BasicEncoder: every convolution is followed by
 * norm_fn = "instance"): mean_rstd [n][Cout][2] = (mean, 1 / sqrt(biased var + eps)) of y over H x W, summed from the convolution's
 * registers per 16 x 16 tile (fp32 over 64 pixels, fp64 beyond).  ws: n * ceil(H / 16) * ceil(W / 16) * Cout * 16 bytes. */
int sampt_conv3x3_planes_instnorm_stats(const void* x_hl, const void* w_hl, const float* bias, float* y, int n, int H, int W, int Cin,
                                        int Cout, float eps, float* mean_rstd, void* ws, size_t ws_bytes, sampt_stream_t stream);
/* The mask decoder's image-side projection as an operator: C [M][N] = act(A W^T + bias) + res[row % res_mod (0: row)] with f32 A
 * [M][K] (K % 32 == 0) and W as split-fp16 planes [2][N][K] scaled by 2^8 (pack.split_f16x3) — 3-term fp16 MFMA products, fp32-grade
 * (reference: segment_anything/modeling/transformer.py:185-232 q / k / v / out projections over the image tokens).  shuf_g > 0: the
 * ConvTranspose2d(k = 2, s = 2) form (mask_decoder.py:53-61): N = 4 * cout columns ordered (dy, dx, channel), GEMM row f g^2 + y g + x
 * goes to pixel row f 4 g^2 + (2 y + dy) 2 g + 2 x + dx of C [.][cout]; bias has cout entries; res must be null.  act: 0 none, 1 ReLU,
 * 2 GELU (erf). */
int sampt_gemm_x3_rows(const float* A, const void* w_hl, const float* bias, const float* res, int res_mod, float* C, int M, int N,
                       int K, int act, int shuf_g, sampt_stream_t stream);
/* sampt_gemm_x3_rows with the NEXT operator of the mask decoder's output_upscaling (mask_decoder.py:53-61, 118-126) done on the rows
 * while they are in registers (weights-resident kernel only: M >= 16384, shuf_g > 0, act = 2, res null — anything else is refused):
 *   epi = 1 (K = 256, N = 4 x 64): LayerNorm2d over the 64 channels of every output pixel (epi_a / epi_b = weight / bias [64],
 *            epi_eps), then GELU: C as sampt_gemm_x3_rows;
 *   epi = 2 (K = 64, N = 4 x 32): GELU, then <pixel's 32 channels, epi_a[frame * epi_ld + 0 .. 31]> with frame = row / shuf_g^2:
 *            C [M * 4] holds one float per output pixel (the low-resolution mask logits of predict_masks);
 *   epi = 3 (K = 128, N = 256, shuf_g = 0, act = 0, res [M][256] non-null): C = LayerNorm(res + A W^T + bias) over the row, weights
 *            epi_a / epi_b [256] (transformer.py:145-150: keys = norm4(keys + attn_out)); C may be res (in place).
 * Bit for bit what sampt_gemm_x3_rows followed by sampt_layernorm (D = 64, act 2 / D = 256) / sampt_sam_mask_dot computes. */
int sampt_gemm_x3_rows_epi(const float* A, const void* w_hl, const float* bias, const float* res, int res_mod, float* C, int M, int N,
                           int K, int act, int shuf_g, int epi, const float* epi_a, const float* epi_b, float epi_eps, int epi_ld,
                           sampt_stream_t stream);
/* low [frames][npix] = <up [frames][npix][C], hyper [frames][ld_hyper]> over C channels (MaskDecoder.predict_masks: hyper_in @ upscaled). */
int sampt_sam_mask_dot(const float* up, const float* hyper, int ld_hyper, float* low, int frames, int npix, int C, sampt_stream_t stream);
/* Process-wide A / B switch: 1 (default) = 1 x 1 split-fp16 "convolutions" over f32 activations with M >= 16384 rows and K = 64 /
 * 128 / 256 (the mask decoder's image-side projections and transposed convolutions) run on the weights-resident-in-LDS kernel of
 * csrc/gemm_x3_wres.hip; 0 = on the tiled kernel of rounds 2 - 5 (csrc/conv_f16x3.hip k_conv_f16x3); 2 = weights-resident, but the
 * decoder's LayerNorm2d + GELU and mask dot product stay kernels of their own.  Results are bitwise identical in all three. */
int sampt_gemm_set_wres(int on);
/* Process-wide knob of the PIPS window's MLP-Mixer (csrc/pips_mixer.hip).  fused = 1 (default): two launches per mixer block —
 * [sum of the previous channel MLP's slabs + residual -> token mixing] and [LayerNorm -> fc1 -> GELU -> fc2 over hidden slices];
 * fused = 0: the four-launch blocks of rounds 1 - 5 (token mixing, LayerNorm, two thin GEMMs); fused = 2: the two-launch blocks with
 * the channel MLP as 3-term split-fp16 MFMA products at fp32 grade (csrc/pips_mixer_x3.hip; needs the packed operand streams
 * "delta_block.to_delta.<i>.__x3s16" / "__x3s32" among the weights, else it falls back to fused = 1).  workgroups = how many
 * workgroups a channel-MLP launch should reach (it uses 8, 16 or 32 hidden slices per group of two point chains; default 32 =
 * the CUs the ViT encoder's persistent GEMMs leave free beside the tracker). */
int sampt_pips_set_mixer(int fused, int workgroups);
/* A HIP stream confined to the CUs [cu_lo, cu_hi) of EVERY XCD (hipExtStreamCreateWithCUMask; MI355X: 8 XCDs of 32 CUs, mask bit i
 * = CU i / 8 of XCD i % 8 — tools/probes/cu_mask_probe.hip, profiles/r6_c3_cu_mask_probe.log; an XCD left without any CU gets all of
 * them back, so every XCD keeps at least one).  The host orchestration uses two of them to split the chip in space between the ViT
 * encoder and the latency-bound side work (tracker window rounds, decoder chains): kernels of one never wait for CUs of the other.
 * priority: 0 normal, -1 high (the runtime's mask call has no priority argument: recorded for the caller, currently unused).
 * Destroy with sampt_stream_destroy after the stream is idle. */
int sampt_stream_create_cu_range(int cu_lo, int cu_hi, sampt_stream_t* out);
int sampt_stream_destroy(sampt_stream_t stream);
/* Calibration hook of the fp16 mode's static bias correction (sam_pt_amd/sam_predictor.py: the rounding of a weight matrix to
 * fp16 adds A.(W - fp16(W))^T to a GEMM's output; its token-mean part mean(A).(W - fp16(W))^T is a per-column constant that the
 * packer folds into the bias once per frame geometry).  While colmeans_dev is set, every block GEMM of sampt_vit_encode (fp16 mode,
 * plain entry point, one frame) also writes the column means of its A operand over the frame's real tokens to
 * colmeans_dev[(block * 4 + kind) * ld + k], kind = 0 qkv, 1 proj, 2 fc1, 3 fc2; ld >= mlp_ratio * embed_dim.  NULL ends it. */
int sampt_vit_calibrate(sampt_vit_t h, float* colmeans_dev, int ld);
/* Measurement hook: between begin and end every fp16 GEMM launch of sampt_vit_encode is bracketed by HIP events on the
 * launching stream; end waits for them and returns the summed algorithmic FLOP (2*M*N*K), the summed kernel time and the
 * number of launches — the in-situ figures behind bench.py's `roofline`. */
int sampt_vit_profile_begin(sampt_vit_t h);
int sampt_vit_profile_end(sampt_vit_t h, double* flop, double* ms, int* launches);
/* frames_dev: uint8, (B,3,H,W) if chw else (B,H,W,3), H,W <= img_size with the longest side == img_size already
 * (the reference pipelines resize before SamPt: configs/demo.yaml:20, configs/vos_eval_root.yaml:28).
 * features_dev: float32 [B][grid*grid][out_chans] — the (B,256,64,64) embedding in NHWC / token-major order.
 * interm_out_dev: NULL, or float32 [B][grid*grid][embed_dim] receiving the token stream after the first
 * global-attention block — HQ-SAM's interm_embeddings[0] (configs/model/sam/samhq_vit_*.yaml, MaskDecoderHQ). */
int sampt_vit_encode(sampt_vit_t h, const uint8_t* frames_dev, int chw, int B, int H, int W, float* features_dev,
                     float* interm_out_dev, void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
/* The same result (bit for bit) with the frame-independent part of the work done once per frame geometry.  Sam.preprocess
 * (sam.py: normalise, then zero-pad to img_size^2) leaves the token rows below a landscape frame — 28 of 64 rows for 16:9
 * video — without any pixel; until the first global-attention block such a token only ever meets tokens of its own
 * window, so whole window rows of them evolve identically in every frame.
 *   sampt_vit_live_rows: *live_rows = the token rows that can depend on the frame before the first global block
 *     (ceil(H/patch) completed to whole windows), == grid when nothing can be skipped (portrait / square frames, a
 *     global block first); *cache_bytes = size of the dead-row cache [(grid - live_rows) * grid][embed_dim] f32.
 *   sampt_vit_encode_live(build=1): fills dead_cache_dev from ONE frame of that geometry (frames_dev[0]; features_dev /
 *     interm_out_dev unused).  build=0: encodes B frames, running the blocks before the first global one on the live
 *     rows only and taking the other rows from dead_cache_dev.  Workspace: sampt_vit_encode_workspace_bytes. */
int sampt_vit_live_rows(sampt_vit_t h, int H, int W, int* live_rows, size_t* cache_bytes);
int sampt_vit_encode_live(sampt_vit_t h, const uint8_t* frames_dev, int chw, int B, int H, int W, float* features_dev,
                          float* interm_out_dev, float* dead_cache_dev, int build, void* workspace_dev,
                          size_t workspace_bytes, sampt_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * seam 2b — prompt encoder + mask decoder + postprocess = SamPredictor.predict_torch(multimask_output=False,
 * return_logits=True) as called at sam_pt.py:783-828.
 * --------------------------------------------------------------------------------------------------------- */
/* max_frames: capacity of the frame batch of sampt_sam_track_decode (the packed pixel-shuffle row maps
 * "mask_decoder.__up0_map" / "__up1_map" must cover it, see sam_pt_amd/pack.py).
 * vit_dim: 0 = SAM MaskDecoder (configs/model/sam/mask_decoder/sam.yaml); > 0 = HQ-SAM MaskDecoderHQ
 * (configs/model/sam/samhq_vit_huge.yaml:8-10, vit_dim = image encoder embed_dim), which needs the extra
 * "mask_decoder.{hf_token,hf_mlp,compress_vit_feat,embedding_encoder,embedding_maskfeature}" weights. */
int sampt_dec_create(const char* const* names, const void* const* ptrs, int n, int grid, int img_size, int max_frames,
                     int vit_dim, sampt_dec_t* out);
void sampt_dec_destroy(sampt_dec_t h);
/* Workspace for prompts of up to 120 points (every shipped SAM-PT config with <= 7 objects per batch); for larger
 * prompts (k <= SAMPT_DEC_MAX_POINTS: the reference accepts any k — 16 points x M objects fed to each other as
 * negatives, sam_pt.py:737-756, or the VIS adapter's 100-mask batches) size it with the _k variant. */
#define SAMPT_DEC_MAX_POINTS 4000
int sampt_dec_workspace_bytes(sampt_dec_t h, int frames, int out_h, int out_w, size_t* bytes);
int sampt_dec_workspace_bytes_k(sampt_dec_t h, int frames, int k, int out_h, int out_w, size_t* bytes);
/* HQ-SAM only: hq_features_dev [frames][16*grid*grid][32] = embedding_encoder(features) +
 * compress_vit_feat(interm) (MaskDecoderHQ.forward), computed once per frame from sampt_vit_encode's two outputs
 * and passed to every decode pass of that frame. */
int sampt_dec_hq_workspace_bytes(sampt_dec_t h, int frames, size_t* bytes);
int sampt_dec_hq_features(sampt_dec_t h, int frames, const float* features_dev, const float* interm_dev,
                          float* hq_features_dev, void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
/* One pass, one frame.  features_dev [grid*grid][256]; hq_features_dev: sampt_dec_hq_features output for the frame
 * (HQ-SAM handles) or NULL (SAM handles); pts_dev [k][2] (input-frame pixels), labels_dev int32 [k];
 * box_dev 4 floats or NULL; mask_in_dev [4*grid][4*grid] low-res logits or NULL.  Limit: k <= SAMPT_DEC_MAX_POINTS prompt
 * points (tokens-as-keys attention is tiled over the tokens; SAMPT_ERR_UNSUPPORTED beyond), workspace sized for k.  Outputs: logits_out_dev
 * [out_h][out_w], iou_out_dev [1], low_res_out_dev [4*grid][4*grid]. */
int sampt_sam_decode(sampt_dec_t h, const float* features_dev, const float* hq_features_dev, const float* pts_dev,
                     const int32_t* labels_dev, int k, const float* box_dev, const float* mask_in_dev, int in_h, int in_w, int out_h, int out_w,
                     float* logits_out_dev, float* iou_out_dev, float* low_res_out_dev, void* workspace_dev,
                     size_t workspace_bytes, sampt_stream_t stream);
/* The same pass with multimask_output=True (SAM decoder only): the 3 masks / IoU predictions of mask tokens 1..3, in
 * that order (MaskDecoder.forward mask_slice = slice(1, None)).  logits_out_dev [3][out_h][out_w], iou_out_dev [3],
 * low_res_out_dev [3][4*grid][4*grid].  Not on the SAM-PT path (sam_pt.py always passes False); part of the
 * SamPredictor.predict_torch surface. */
int sampt_sam_decode_multimask(sampt_dec_t h, const float* features_dev, const float* pts_dev, const int32_t* labels_dev,
                               int k, const float* box_dev, const float* mask_in_dev, int in_h, int in_w, int out_h,
                               int out_w, float* logits_out_dev, float* iou_out_dev, float* low_res_out_dev,
                               void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
/* Whole SamPt.predict_mask chain (sam_pt.py:760-837) for `frames` independent (frame, object) items that share the
 * visible-point count k, batched into one launch sequence and without host synchronisation:
 * [positives-only pass over the first n_pos_first points when n_pos_first >= 0, i.e. negative_points_per_mask > 0;
 * pass -1 for the single-pass case] -> all-points pass -> `refine_iters` box+mask refinement passes
 * (bbox of logits>0 and the `sum < 2 -> stop` rule evaluated per item on device) -> logits = -inf if iou < iou_thr.
 * Ragged batches: k_item_dev / npos_item_dev (int32 [frames], or NULL for uniform) give each item's own point count
 * (<= k) and leading-positive count (<= n_pos_first); an item's prompt tokens are packed in front of its token matrix
 * and the padding rows are masked out of every attention, so each item's result equals its un-batched one.
 * features_dev [frames][grid*grid][256]; hq_features_dev [frames][16*grid*grid][32] (HQ-SAM) or NULL;
 * pts_dev [frames][ld_pts][2]; labels_dev int32 [frames][ld_pts];
 * final_logits_dev [frames][out_h][out_w]; score_out_dev [frames] = predicted IoU. */
int sampt_sam_track_decode(sampt_dec_t h, int frames, const float* features_dev, const float* hq_features_dev,
                           const float* pts_dev, const int32_t* labels_dev, int k, const int32_t* k_item_dev,
                           const int32_t* npos_item_dev, int ld_pts, int n_pos_first, int refine_iters, float iou_thr,
                           int in_h, int in_w, int out_h, int out_w, float* final_logits_dev, float* score_out_dev,
                           void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
/* The same chain replayed from a hipGraph.  The launch sequence of sampt_sam_track_decode depends only on its arguments, so
 * it is stream-captured once per distinct argument tuple — every scalar AND every pointer: the caller keeps the inputs,
 * outputs and workspace of a prompt bucket in persistent buffers (sam_pt_amd.SamPredictor.track_decode(graph=True)) — and
 * replayed with one hipGraphLaunch afterwards.  First call of a tuple: plain launches; second: capture + instantiate +
 * launch; then: launch only.  At most 64 instantiated graphs are cached per handle (least recently used one dropped).
 * `stream` must not be the NULL stream (not capturable: falls back to plain launches).  Results are those of
 * sampt_sam_track_decode bit for bit (same kernels, same order).  sampt_dec_graph_stats: cache size / captures / replays. */
int sampt_sam_track_decode_graph(sampt_dec_t h, int frames, const float* features_dev, const float* hq_features_dev,
                                 const float* pts_dev, const int32_t* labels_dev, int k, const int32_t* k_item_dev,
                                 const int32_t* npos_item_dev, int ld_pts, int n_pos_first, int refine_iters, float iou_thr,
                                 int in_h, int in_w, int out_h, int out_w, float* final_logits_dev, float* score_out_dev,
                                 void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
int sampt_dec_graph_stats(sampt_dec_t h, long* cached, long* captures, long* launches);
int sampt_postprocess_masks(const float* low_res_dev, int L, int img_size, int in_h, int in_w, float* out_dev, int out_h,
                            int out_w, sampt_stream_t stream);
/* bbox_state_dev: int32[5] = {xmin, ymin, xmax, ymax, count} over logits > 0 (sam_pt.py:809-820). */
size_t sampt_bbox_workspace_bytes(int h, int w);
int sampt_bbox_from_logits(const float* logits_dev, int h, int w, int32_t* bbox_state_dev, void* workspace_dev,
                           size_t workspace_bytes, sampt_stream_t stream);

/* One axis of PIL's 8-bit separable resampler (Image.resize, the arithmetic behind ResizeLongestSide.apply_image =
 * torchvision resize(to_pil_image(.)) in SamPredictor.set_image, App. A-1): src_dev [outer][in_len][inner] uint8 ->
 * dst_dev [outer][out_len][inner], dst = clip8((2^21 + sum_x src[xmin+x]*coef[xx][x]) >> 22).  coef_dev int32
 * [out_len][ksize] / bounds_dev int32 [out_len][2] = PIL's precompute_coeffs + normalize_coeffs_8bpc tables, built by the
 * host (sam_pt_amd/sam_predictor.py); horizontal pass first, then vertical, each rounded to uint8 as PIL does. */
int sampt_pil_resample_u8(const uint8_t* src_dev, uint8_t* dst_dev, long outer, int in_len, int out_len, int inner,
                          const int32_t* coef_dev, const int32_t* bounds_dev, int ksize, sampt_stream_t stream);

/* VOS post-processing.  sampt_resize_logits: F.interpolate(logits, target_hw, bilinear, align_corners=False) of
 * sam_pt.py:205-206 for n single-channel maps.  sampt_index_masks: uint8 object index per pixel = argmax over
 * {background logit 0, logits_dev[M][npix]}, i.e. the bg-stack + softmax + argmax of vos_eval/eval.py:304, 326, 355. */
int sampt_resize_logits(const float* src_dev, int n, int sh, int sw, float* dst_dev, int dh, int dw, sampt_stream_t stream);
int sampt_index_masks(const float* logits_dev, int M, long npix, uint8_t* out_dev, sampt_stream_t stream);
/* The same for logits_dev [M][T][hw] with the evaluator's overrides (vos_eval/eval.py:318-326): object m is -1e8 on
 * frames before query_t_dev[m] (int32 [M]) and, when gt_masks_dev (uint8 [M][hw], already at this resolution) is given,
 * +-1e8 from its ground-truth mask on the query frame.  out_dev uint8 [T][hw]. */
int sampt_vos_index_masks(const float* logits_dev, int M, int T, long hw, const int32_t* query_t_dev,
                          const uint8_t* gt_masks_dev, uint8_t* out_dev, sampt_stream_t stream);
/* ... and for frames processed at (h, w) but wanted at (out_h, out_w) (eval.py:326, 340-356: softmax, bilinear resize of
 * the probabilities with align_corners=False, argmax).  logits_dev [M][T][h][w], gt_masks_dev [M][h][w] or NULL,
 * out_dev uint8 [T][out_h][out_w]; M <= 32. */
int sampt_vos_index_masks_resized(const float* logits_dev, int M, int T, int h, int w, const int32_t* query_t_dev,
                                  const uint8_t* gt_masks_dev, int out_h, int out_w, uint8_t* out_dev,
                                  sampt_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Kernel-level entry points (used by the parity tests and the roofline bench; same kernels the engines launch).
 * --------------------------------------------------------------------------------------------------------- */
/* C[M][N] = act(alpha * A[M][K] . W[N][K]^T + bias) (+ res).  dtype: 0 = f32 (exact f32 MFMA), 1 = f16 in / f32 out,
 * 2 = f16 in / f16 out.  act: 0 none, 1 relu, 2 gelu(erf). */
int sampt_gemm(int dtype, const void* A_dev, const void* W_dev, const float* bias_dev, const float* res_dev,
               void* C_dev, int M, int N, int K, int act, float alpha, sampt_stream_t stream);
/* The same GEMM with the row maps the ViT engine uses: rowmap[m] = destination row of C / res for GEMM row m (-1 drops the
 * row; C must hold max(rowmap)+1 rows), a_rowmap[m] = row of A read by GEMM row m (gather), res_mod > 0: the residual row
 * is (destination row) % res_mod (broadcast over the batch, e.g. a positional embedding), ldr = row stride of res (0: N). */
int sampt_gemm_ex(int dtype, const void* A_dev, const void* W_dev, const float* bias_dev, const float* res_dev,
                  void* C_dev, int M, int N, int K, int act, float alpha, const int32_t* rowmap_dev,
                  const int32_t* a_rowmap_dev, int res_mod, int ldr, sampt_stream_t stream);
/* dtype 3 / 4 of sampt_gemm_ex: fp32-grade product on the fp16 matrix pipe.  A [M][2K] and W [N][2K] are "x3 rows": every
 * block of 32 consecutive k of a logical row is stored as 64 halves hi(32) | lo(32), v = hi + lo with hi = fp16(v),
 * lo = fp16(v - hi) (sampt_split_rows_x3; weights are split after scaling by 2^8 — sam_pt_amd/pack.py:x3_rows — and the
 * caller passes alpha = 2^-8).  K is the LOGICAL K (K % 64 == 0).  dtype 3: C f32 [M][N]; dtype 4: C x3 rows [M][2N]. */
int sampt_split_rows_x3(const float* x_dev, void* y_dev, int M, int K, sampt_stream_t stream);
/* NHWC convolution as implicit GEMM: x [n][H][W][Cin], w [Cout][KH][KW][Cin], y [n][OH][OW][Cout] (f32 out).
 * dtype 3 = fp32-grade result on the fp16 matrix pipe: x is f32, w_dev is half [2][Cout][KH*KW*Cin] = the weights times
 * 2^8 split as hi = fp16(w), lo = fp16(w - hi) (sam_pt_amd/pack.py:split_f16x3); Cin % 32 == 0, Cout % 4 == 0.
 * dtype 4 = the same with PRE-SPLIT activations: x_dev is half [2][n][H][W][Cin] = hi plane fp16(x), lo plane fp16(x - hi)
 * (what the tracker encoder's InstanceNorm writes); all four operand planes then reach LDS by LDS-DMA. */
int sampt_conv2d_nhwc(int dtype, const void* x_dev, const void* w_dev, const float* bias_dev, float* y_dev, int n,
                      int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, sampt_stream_t stream);
int sampt_instance_norm_nhwc(float* x_dev, int n, int hw, int C, float eps, int relu, const float* skip_dev,
                             void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
size_t sampt_instance_norm_workspace_bytes(int n, int hw, int C);
/* out_f16: 0 = f32 rows, 1 = fp16 rows, 2 = x3 rows [M][2D] (see sampt_split_rows_x3; D % 32 == 0) */
int sampt_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, void* y_dev, int M, int D, float eps,
                    int out_f16, int act, sampt_stream_t stream);
int sampt_resize_bilinear_nhwc(const float* src_dev, int n, int sh, int sw, int C, float* dst_dev, int dh, int dw,
                               int dstC, int c_off, int align_corners, sampt_stream_t stream);
int sampt_avgpool2x2_nhwc(const float* src_dev, int n, int h, int w, int C, float* dst_dev, sampt_stream_t stream);
/* Fused correlation + 7x7x4-level window sampler (CorrBlock.corr + .sample, pips.py:364-407).
 * ffeats_dev [n][S][128]; coords_dev [S][n][2] (level-0 feature-map pixels); out_dev [n][S][196]. */
int sampt_corr_sample_f32(const float* const pyr_dev[4], int H0, int W0, const int32_t* frame_idx_dev, int S, int n,
                          const float* ffeats_dev, const float* coords_dev, float* out_dev, sampt_stream_t stream);
/* The two kernels of a fused MLP-Mixer block (csrc/pips_mixer.hip; pips.py:96-128), exported for the kernel tests.
 * mlp: part_dev[slice][nseq*8][512] = fc2 over the slice's hidden units of gelu(fc1(LayerNorm(x)) + b1); x_dev [nseq*8][512],
 *      w1 [2048][512], b1 [2048], w2 [512][2048]; slices = 8, 16 or 32 (what sampt_pips_set_mixer's workgroup target selects).
 * reduce: x' = res + (sum of the `slices` slabs in order + bias) (slices = 0: x' = res, part / bias unused);
 *      mode 0: out_dev [nseq*8][512] = x' + token-mix(LayerNorm(x')) with tw1 [32][8], tb1 [32], tw2 [8][32], tb2 [8];
 *      mode 1: out_dev [nseq][512] = mean over the 8 tokens of LayerNorm(x') (token weights unused).  out_dev != res_dev. */
int sampt_pips_mix_mlp_f32(const float* x_dev, const float* lnw_dev, const float* lnb_dev, const float* w1_dev,
                           const float* b1_dev, const float* w2_dev, float* part_dev, int nseq, int slices,
                           sampt_stream_t stream);
int sampt_pips_mix_reduce_f32(const float* part_dev, int slices, const float* bias_dev, const float* res_dev, int nseq, int mode,
                              const float* lnw_dev, const float* lnb_dev, const float* tw1_dev, const float* tb1_dev,
                              const float* tw2_dev, const float* tb2_dev, float* out_dev, sampt_stream_t stream);
/* The split-fp16 generation of the same block (csrc/pips_mixer_x3.hip), exported for the kernel tests.
 * pre: x' = res + (slab sum + bias) (slices = 0: x' = res); xout_dev [nseq*8][512] = x' + token-mix(LayerNorm1(x')); xop_dev = the
 *      operand images of 2^6 LayerNorm2(xout) split into fp16 hi / lo, sampt_pips_mix_xop_halves(nseq) halves.
 * mlp_x3: part_dev[slice][nseq*8][512] from xop_dev and the block's packed weight stream wstream_dev [slices][(2048/slices/16)*64*512]
 *      halves (sam_pt_amd/pack.py pips_mixer_x3_stream; slices = 16 or 32) and the fc1 bias b1_dev [2048]. */
size_t sampt_pips_mix_xop_halves(int nseq);
int sampt_pips_mix_pre_f32(const float* part_dev, int slices, const float* bias_dev, const float* res_dev, int nseq,
                           const float* ln1w_dev, const float* ln1b_dev, const float* tw1_dev, const float* tb1_dev,
                           const float* tw2_dev, const float* tb2_dev, const float* ln2w_dev, const float* ln2b_dev,
                           float* xout_dev, void* xop_dev, sampt_stream_t stream);
int sampt_pips_mix_mlp_x3(const void* xop_dev, const void* wstream_dev, const float* b1_dev, float* part_dev, int nseq, int slices,
                          sampt_stream_t stream);
/* ViT attention on a packed qkv matrix [B*S*S][3*heads*hd] (f16), decomposed rel-pos tables (2S-1, hd) f32.
 * out_dev f16 [B*S*S][heads*hd].  The bias tables are built inside the kernel; the workspace arguments are kept for
 * ABI stability and ignored (may be NULL / 0). */
/* The mask decoder's two small fp32 attention kernels: q [F][Nq][heads*hd], k / v [F][Nk][heads*hd], out like q;
 * nk_item_dev (int32 [F] or NULL): per-item number of valid keys of a ragged batch.  kind 0 = few queries against up to
 * 4096 keys: hd 32 (token self-attention) runs wave = (64 queries, head) with scalar-loaded key rows split over 4 waves,
 * hd 16 one workgroup per (4 queries, head, item) — the engine's token->image attention is sampt_attention_t2i_f32 below;
 * kind 1 = many queries against few keys (image->token, hd 16, any Nk): with 8 heads wave = head, lane = query and the
 * K / V rows are scalar loads, otherwise one thread per (query, head) with keys staged through LDS in chunks of 128. */
int sampt_attention_f32(int kind, const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, int F, int Nq,
                        int Nk, int heads, int hd, const int32_t* nk_item_dev, sampt_stream_t stream);
/* The decoder's token -> image attention (8 heads x 16 channels: q [F][Nq][128], k / v [F][Nk][128]) as the engine runs
 * it: the keys of a frame are split over workgroups that each read their K / V rows once, fully coalesced, for all heads
 * and up to 8 queries (running softmax), and a second launch merges the splits.  Same result as kind 0 above up to fp32
 * re-association. */
int sampt_attention_t2i_workspace_bytes(int F, int Nq, int Nk, size_t* bytes);
int sampt_attention_t2i_f32(const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, int F, int Nq, int Nk,
                            void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
/* CoTracker's UpdateFormer attention straight from packed qkv rows [rows][3*heads*hd] (timm Attention): token t of group b
 * is row b*batch_stride_rows + t*token_stride_rows (time attention: S, 1; space attention: 1, S); out [rows][heads*hd]. */
int sampt_cotracker_attention_f32(const float* qkv_dev, float* out_dev, int nbatch, int L, int batch_stride_rows,
                                  int token_stride_rows, int heads, int hd, sampt_stream_t stream);
/* SAM's ViT attention (sam/modeling/image_encoder.py Attention.forward with use_rel_pos: third-party segment-anything, absent
 * from /root/reference; restated in oracle/sam_ref.py) for B windows (or whole frames) of S x S tokens straight from the
 * packed fp16 qkv rows [B*S*S][3*heads*hd]; rel_h / rel_w: the block's rel_pos tables f32 [2S-1][hd] (the decomposed bias
 * is computed inside the kernel); out fp16 [B*S*S][heads*hd].  Supported geometries: (S, hd) = (64, 80), (64, 64),
 * (14, 80), (14, 64), (16, 32), (6, 32); anything else returns SAMPT_ERR_UNSUPPORTED.  workspace_dev / workspace_bytes
 * are unused (kept for ABI stability: K / V tiles are staged by LDS-DMA, nothing goes through HBM scratch). */
int sampt_vit_attention_f16(const void* qkv_dev, const float* rel_h_dev, const float* rel_w_dev, void* out_dev, int B,
                            int S, int heads, int hd, void* workspace_dev, size_t workspace_bytes, sampt_stream_t stream);
/* K-medoids query-point selection (SamPt in query_masks mode / point re-initialisation: sam_pt/utils/query_points.py:62-99,
 * third-party sklearn_extra KMedoids(method="alternate", init="heuristic") restated in sam_pt_amd/query_points.py) on the
 * device, bit-identical to that host restatement: fp64 distances, numpy's pairwise summation order, first-index ties.
 * xy_dev: [n][2] f32 pixel coordinates (integers), n <= 2048.
 *   rowsums:   out_dev[i] = sum_j |xy_i - xy_j|  (the heuristic initialisation partitions these on the host: np.argpartition)
 *   alternate: medoids_dev int32 [K] holds the initial medoids on entry and the converged ones on return (K <= 64);
 *              iters_dev (int32, may be NULL) receives the number of iterations run (<= max_iter). */
int sampt_kmedoids_rowsums_f64(const float* xy_dev, int n, double* out_dev, sampt_stream_t stream);
int sampt_kmedoids_alternate(const float* xy_dev, int n, int K, int32_t* medoids_dev, int max_iter, int32_t* iters_dev,
                             sampt_stream_t stream);
/* Shi-Tomasi query points and mask erosion on the device (replaces the cv2 calls of
 * sam_pt/utils/query_points.py:102-162 extract_corner_points and :165-194 erode_mask_proportional_to_its_furthest_points_distance;
 * bit-identical to the numpy restatement of OpenCV's algorithms in sam_pt_amd/query_points.py).
 *   erode:      out = cv2.erode(mask, ones((k, k))) of a {0, 1} byte mask [H][W]; k = 0 means the default 3 x 3, k = 1 a copy;
 *               tmp_dev: H*W scratch bytes.
 *   shi_tomasi: image_dev u8 [3][H][W] RGB, mask_dev u8 {0, 1} [H][W].  Gray conversion, the 6 % / 2 % / 1 % / no erosion cascade
 *               (>= 10 surviving pixels), cornerMinEigenVal(3, 3), threshold at quality_level x the maximum inside the eroded
 *               mask, 3 x 3 local maxima, greedy selection of up to n_points (<= 64) corners at minDistance = the eroded mask's
 *               bounding-box diagonal / n_points — one launch sequence without a host round trip.  out_xy_dev f32 [n_points][2]
 *               (x, y); out_info_dev int32 [16]: [12] corners found, [10] k of the erosion kept (-1: the mask itself), [9] pixels
 *               of the eroded mask, [5..8] its bounding box (ymin, ymax, xmin, xmax), [0..4] the mask's bounding box and count. */
int sampt_qp_corners_workspace_bytes(int H, int W, size_t* bytes);
int sampt_qp_erode_u8(const uint8_t* mask_dev, int H, int W, int k, uint8_t* tmp_dev, uint8_t* out_dev, sampt_stream_t stream);
int sampt_qp_shi_tomasi(const uint8_t* image_dev, const uint8_t* mask_dev, int H, int W, int n_points, float quality_level,
                        float* out_xy_dev, int32_t* out_info_dev, void* ws_dev, size_t ws_bytes, sampt_stream_t stream);
/* The same attention at fp32 grade (precision "f16x3": three split-fp16 MFMAs per product in Q.K^T, the bias tables and
 * P.V; fp32 softmax).  qkv_dev: x3 rows [B*S*S][2*3*heads*hd] halves (what the dtype-4 qkv GEMM writes), out_dev: x3 rows
 * [B*S*S][2*heads*hd].  Same geometries as sampt_vit_attention_f16; heads*hd % 32 == 0. */
int sampt_vit_attention_x3(const void* qkv_dev, const float* rel_h_dev, const float* rel_w_dev, void* out_dev, int B, int S,
                           int heads, int hd, sampt_stream_t stream);
/* The windowed blocks as the encoder runs them (SAM's window_partition, App. A-3): qkv_dev holds frames x nwy x nwx windows of
 * S x S tokens in window order; token (iy, ix) of window (wy, wx) is zero PADDING when wy*S + iy >= grid_h or wx*S + ix >=
 * grid_w.  SAM pads after norm1, so a padded token's qkv is the bias alone: the kernels take its K / V from bias_row_dev (the
 * qkv bias as ONE row in the matrix's format: 3*heads*hd halves for precision 1, an x3 row of twice that for precision 2) and
 * never read the qkv rows of padded tokens (they may be uninitialised; the output rows of padded queries are meaningless). */
int sampt_vit_window_attention(int precision, const void* qkv_dev, const float* rel_h_dev, const float* rel_w_dev, void* out_dev,
                               int frames, int S, int heads, int hd, const void* bias_row_dev, int nwx, int nwy, int grid_h,
                               int grid_w, sampt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMPT_HIP_H */
