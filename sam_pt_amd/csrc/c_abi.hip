// extern "C" surface of libsampt_hip.so (declared in include/sampt_hip.h).
#include <string.h>

#include <exception>
#include <memory>
#include <string>
#include <unordered_map>

#include "../../include/sampt_hip.h"
#include "engine.h"

namespace sampt {
static thread_local std::string g_err;
void set_error(const char* where, hipError_t e) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
}
const char* last_error() { return g_err.c_str(); }
static int fail(int rc, const std::string& msg) {
  g_err = msg;
  return rc;
}
static WeightMap make_map(const char* const* names, const void* const* ptrs, int n) {
  WeightMap w;
  for (int i = 0; i < n; ++i) w.m[names[i]] = ptrs[i];
  return w;
}
// Builds a handle; no C++ exception (allocation failure, ...) may cross the C boundary.
template <class H, class Init>
static int create_handle(const char* what, H** out, Init init) {
  try {
    std::unique_ptr<H> h(new H());
    int rc = init(*h);
    if (rc != SAMPT_OK) return fail(rc, h->e.error.empty() ? std::string(what) + ": initialisation failed" : h->e.error);
    *out = h.release();
    return SAMPT_OK;
  } catch (const std::exception& ex) {
    return fail(SAMPT_ERR_ARG, std::string(what) + ": " + ex.what());
  } catch (...) {
    return fail(SAMPT_ERR_ARG, std::string(what) + ": unknown C++ exception");
  }
}
}  // namespace sampt

using namespace sampt;

struct sampt_pips {
  PipsEngine e;
  int* flag = nullptr;                       // pinned host int32[2]: active chains after the last two rounds (sampt_pips_track_f32)
  hipEvent_t flag_ev[2] = {nullptr, nullptr};
  ~sampt_pips() {
    if (flag) (void)hipHostFree(flag);
    for (auto& ev : flag_ev)
      if (ev) (void)hipEventDestroy(ev);
  }
};
struct sampt_pips2 { Pips2Engine e; };
struct sampt_cotracker { CotEngine e; };
struct sampt_vit { VitEngine e; };
// hipGraph cache of the per-(frame, object) decode chain (north_star: "hipGraph capture of the per-frame decode"): one
// instantiated graph per distinct call signature (every scalar AND every pointer of sampt_sam_track_decode_graph).
struct DecGraphKey {
  const void *features, *hq, *pts, *labels, *k_item, *npos_item, *logits, *score, *ws;
  size_t ws_bytes;
  int frames, k, ld_pts, n_pos_first, refine, in_h, in_w, oh, ow;
  float iou_thr;
  bool operator==(const DecGraphKey& o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
};
struct DecGraphKeyHash {
  size_t operator()(const DecGraphKey& k) const {
    const unsigned char* p = (const unsigned char*)&k;
    size_t h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(k); ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
  }
};
struct DecGraphEntry {
  hipGraphExec_t exec = nullptr;
  int seen = 0;       // calls with this signature so far (the first one runs eagerly: lazy one-time kernel attributes)
  long last_use = 0;
};
struct sampt_dec {
  DecEngine e;
  std::unordered_map<DecGraphKey, DecGraphEntry, DecGraphKeyHash> graphs;
  long graph_clock = 0, graph_launches = 0, graph_captures = 0;
  ~sampt_dec() {
    for (auto& kv : graphs)
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  }
};

extern "C" {

int sampt_version(void) { return 1; }
const char* sampt_last_error(void) { return last_error(); }

// ------------------------------------------------------------------------------------------- PIPS
int sampt_pips_create(const char* const* names, const void* const* ptrs, int n, int stride, int S, sampt_pips_t* out) {
  if (!names || !ptrs || !out || S != 8) return fail(SAMPT_ERR_ARG, "sampt_pips_create: bad arguments (S must be 8)");
  return create_handle<sampt_pips>("sampt_pips_create", out, [&](sampt_pips& h) {
    h.e.S = S, h.e.stride = stride;
    return h.e.init(make_map(names, ptrs, n));
  });
}
void sampt_pips_destroy(sampt_pips_t h) { delete h; }

int sampt_pips_fnet_workspace_bytes(sampt_pips_t h, int nf, int H, int W, size_t* bytes) {
  if (!h || !bytes) return SAMPT_ERR_ARG;
  Arena a(nullptr, 0);
  float* out[4] = {nullptr, nullptr, nullptr, nullptr};
  int rc = h->e.fnet(nullptr, nf, H, W, out, a, nullptr);
  *bytes = a.peak + 256;
  return rc;
}

int sampt_pips_fnet_f32(sampt_pips_t h, const uint8_t* frames, int nf, int H, int W, float* const pyr[4], void* ws,
                        size_t ws_bytes, sampt_stream_t stream) {
  // any frame size works (feature map = floor(H / stride), pyramid levels floor-halved like avg_pool2d); the coarsest
  // level must keep >= 2 pixels per axis because the sampler normalises by (size - 1) (pips.py:324-326)
  if (!h || !frames || !pyr || !ws || H < 16 * h->e.stride || W < 16 * h->e.stride)
    return fail(SAMPT_ERR_ARG, "sampt_pips_fnet_f32: bad arguments (H, W must be at least 16*stride)");
  Arena a(ws, ws_bytes);
  return h->e.fnet(frames, nf, H, W, pyr, a, (hipStream_t)stream);
}

int sampt_pips_sample_feat_f32(const float* fmap, int H0, int W0, const int32_t* frame_idx, const float* xy, int n,
                               float* out, sampt_stream_t stream) {
  return pips_sample_feat(fmap, H0, W0, 128, (const int*)frame_idx, xy, n, out, (hipStream_t)stream);
}

static PyramidLevels make_pyr(const float* const pyr[4], int H0, int W0) {
  PyramidLevels p;
  int h = H0, w = W0;
  for (int l = 0; l < 4; ++l) {
    p.base[l] = pyr[l], p.H[l] = h, p.W[l] = w;
    h /= 2, w /= 2;
  }
  return p;
}

int sampt_pips_track_workspace_bytes(sampt_pips_t h, int n, size_t* bytes) {
  if (!h || !bytes || n <= 0) return fail(SAMPT_ERR_ARG, "sampt_pips_track_workspace_bytes: bad arguments");
  Arena a(nullptr, 0);
  PyramidLevels p = {};
  int rounds = 0;
  int rc = h->e.track(p, 2, n, nullptr, nullptr, nullptr, nullptr, 0.9f, 6, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                      nullptr, a, nullptr, &rounds);
  *bytes = a.peak + 256;
  return rc;
}

int sampt_pips_track_f32(sampt_pips_t h, const float* const pyr[4], int H0, int W0, int T, int n, const float* q_dev,
                         const uint8_t* flip_dev, const float* q_host, const uint8_t* flip_host, float vis_threshold, int iters,
                         void* const* chunk_events, const int32_t* chunk_lo, const int32_t* chunk_hi, int nchunks,
                         float* traj_dev, float* vis_dev, void* ws, size_t ws_bytes, sampt_stream_t stream, int32_t* rounds) {
  if (!h || !pyr || !q_dev || !flip_dev || !q_host || !flip_host || !traj_dev || !vis_dev || !ws || !rounds || n <= 0 || T < 1 ||
      (nchunks > 0 && (!chunk_events || !chunk_lo || !chunk_hi)))
    return fail(SAMPT_ERR_ARG, "sampt_pips_track_f32: bad arguments");
  for (int i = 0; i < n; ++i)
    if (!(q_host[i * 3] >= 0.f && q_host[i * 3] <= (float)(T - 1)))
      return fail(SAMPT_ERR_ARG, "sampt_pips_track_f32: query frame outside the clip");
  // The pinned termination flags and their events belong to the handle: ONE track call at a time per handle (callers that want
  // concurrent clips create one tracker handle per stream).  Created on first use, all or nothing.
  if (!h->flag) {
    int* flag = nullptr;
    hipEvent_t evs[2] = {nullptr, nullptr};
    bool okay = hipHostMalloc((void**)&flag, 2 * sizeof(int), hipHostMallocDefault) == hipSuccess;
    for (int i = 0; okay && i < 2; ++i) okay = hipEventCreateWithFlags(&evs[i], hipEventDisableTiming) == hipSuccess;
    if (!okay) {
      for (hipEvent_t e : evs)
        if (e) (void)hipEventDestroy(e);
      if (flag) (void)hipHostFree(flag);
      return fail(SAMPT_ERR_HIP, "sampt_pips_track_f32: could not create the pinned flag / events");
    }
    h->flag = flag, h->flag_ev[0] = evs[0], h->flag_ev[1] = evs[1];
  }
  h->flag[0] = h->flag[1] = -1;
  Arena a(ws, ws_bytes);
  int r = 0;
  int rc = h->e.track(make_pyr(pyr, H0, W0), T, n, q_dev, flip_dev, q_host, flip_host, vis_threshold, iters, chunk_events,
                      (const int*)chunk_lo, (const int*)chunk_hi, nchunks, h->flag, h->flag_ev, traj_dev, vis_dev, a,
                      (hipStream_t)stream, &r);
  *rounds = r;
  if (rc != SAMPT_OK && !h->e.error.empty()) return fail(rc, h->e.error);
  return rc;
}

int sampt_pips_round_launches(sampt_pips_t h) { return h && h->e.window_launches ? h->e.window_launches + 2 : 0; }

int sampt_pips_update_workspace_bytes(sampt_pips_t h, int n, size_t* bytes) {
  if (!h || !bytes || n <= 0) return SAMPT_ERR_ARG;
  Arena a(nullptr, 0);
  PyramidLevels p = {};
  int rc = h->e.update(p, nullptr, n, nullptr, nullptr, 6, nullptr, nullptr, a, nullptr);
  *bytes = a.peak + 256;
  return rc;
}

int sampt_pips_update_f32(sampt_pips_t h, const float* const pyr[4], int H0, int W0, const int32_t* frame_idx, int n,
                          const float* xys, const float* feat_init, int iters, float* traj_out, float* vis_out, void* ws,
                          size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !pyr || !frame_idx || !xys || !feat_init || !traj_out || !vis_out || !ws || n <= 0)
    return fail(SAMPT_ERR_ARG, "sampt_pips_update_f32: bad arguments");
  Arena a(ws, ws_bytes);
  return h->e.update(make_pyr(pyr, H0, W0), frame_idx, n, xys, feat_init, iters, traj_out, vis_out, a,
                     (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------- CoTracker
int sampt_cotracker_create(const char* const* names, const void* const* ptrs, int n, int stride, int S,
                           sampt_cotracker_t* out) {
  if (!names || !ptrs || !out || S != 8 || stride != 4)
    return fail(SAMPT_ERR_ARG, "sampt_cotracker_create: bad arguments (the shipped checkpoint is stride 4, window 8)");
  return create_handle<sampt_cotracker>("sampt_cotracker_create", out, [&](sampt_cotracker& h) {
    h.e.S = S, h.e.stride = stride;
    return h.e.init(make_map(names, ptrs, n));
  });
}
void sampt_cotracker_destroy(sampt_cotracker_t h) { delete h; }

int sampt_resize_frames_f32(const void* frames, int src_u8, long planes, int H, int W, float* out, int out_h, int out_w,
                            sampt_stream_t stream) {
  if (!frames || !out || planes <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0)
    return fail(SAMPT_ERR_ARG, "sampt_resize_frames_f32: bad arguments");
  return resize_planes(frames, src_u8, planes, H, W, out, out_h, out_w, (hipStream_t)stream);
}

int sampt_cotracker_fnet_workspace_bytes(sampt_cotracker_t h, int nf, int H, int W, size_t* bytes) {
  if (!h || !bytes) return SAMPT_ERR_ARG;
  Arena a(nullptr, 0);
  float* out[4] = {nullptr, nullptr, nullptr, nullptr};
  int rc = h->e.enc.fnet(nullptr, nf, H, W, out, a, nullptr);
  *bytes = a.peak + 256;
  return rc;
}

int sampt_cotracker_fnet_f32(sampt_cotracker_t h, const float* frames, int nf, int H, int W, float* const pyr[4], void* ws,
                             size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !frames || !pyr || !ws || H < 16 * h->e.stride || W < 16 * h->e.stride)
    return fail(SAMPT_ERR_ARG, "sampt_cotracker_fnet_f32: bad arguments (H, W must be at least 16*stride)");
  Arena a(ws, ws_bytes);
  return h->e.enc.fnet((const uint8_t*)frames, nf, H, W, pyr, a, (hipStream_t)stream);
}

int sampt_cotracker_track_workspace_bytes(sampt_cotracker_t h, int n, size_t* bytes) {
  if (!h || !bytes || n <= 0) return SAMPT_ERR_ARG;
  Arena a(nullptr, 0);
  PyramidLevels p = {};
  int rc = h->e.track(p, 8, nullptr, n, nullptr, nullptr, nullptr, nullptr, nullptr, 6, nullptr, nullptr, a, nullptr);
  *bytes = a.peak + 256;
  return rc;
}

int sampt_cotracker_track_f32(sampt_cotracker_t h, const float* const pyr[4], int H0, int W0, int T,
                              const int32_t* frame_map, int n, const int32_t* query_t_host, const int32_t* query_t,
                              const float* query_xy, const float* pos_x, const float* pos_y, int iters, float* traj_out,
                              float* vis_out, void* ws, size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !pyr || !frame_map || !query_t_host || !query_t || !query_xy || !pos_x || !pos_y || !traj_out || !vis_out ||
      !ws || n <= 0 || T < h->e.S || iters <= 0)
    return fail(SAMPT_ERR_ARG, "sampt_cotracker_track_f32: bad arguments (T must be at least the window length 8)");
  Arena a(ws, ws_bytes);
  int rc = h->e.track(make_pyr(pyr, H0, W0), T, (const int*)frame_map, n, (const int*)query_t_host, (const int*)query_t,
                      query_xy, pos_x, pos_y, iters, traj_out, vis_out, a, (hipStream_t)stream);
  if (rc == SAMPT_ERR_ARG) return fail(rc, "sampt_cotracker_track_f32: query frames must be sorted ascending and inside [0, T)");
  return rc;
}

// ------------------------------------------------------------------------------------------- ViT
int sampt_vit_create(const sampt_vit_config* cfg, const char* const* names, const void* const* ptrs, int n,
                     int win_batches, sampt_vit_t* out) {
  if (!cfg || !names || !ptrs || !out || cfg->depth > 32) return fail(SAMPT_ERR_ARG, "sampt_vit_create: bad arguments");
  VitConfig c;
  c.D = cfg->embed_dim, c.depth = cfg->depth, c.heads = cfg->num_heads, c.grid = cfg->grid, c.window = cfg->window;
  c.patch = cfg->patch, c.out_chans = cfg->out_chans, c.mlp_ratio = cfg->mlp_ratio, c.img = cfg->img_size;
  c.global_mask = cfg->global_mask, c.f16 = cfg->f16;
  for (int i = 0; i < 3; ++i) c.mean[i] = cfg->pixel_mean[i], c.stdv[i] = cfg->pixel_std[i];
  return create_handle<sampt_vit>("sampt_vit_create", out,
                                  [&](sampt_vit& h) { return h.e.init(make_map(names, ptrs, n), c, win_batches); });
}
void sampt_vit_destroy(sampt_vit_t h) { delete h; }

int sampt_vit_set_gemm_workgroups(sampt_vit_t h, int per_xcd) {
  if (!h || per_xcd < 0 || per_xcd > 32) return fail(SAMPT_ERR_ARG, "sampt_vit_set_gemm_workgroups: per_xcd must be 0 .. 32");
  h->e.gemm_wgs = per_xcd;
  return SAMPT_OK;
}

int sampt_vit_set_gemm_workgroups_kind(sampt_vit_t h, int qkv, int proj, int fc1, int fc2) {
  const int v[4] = {qkv, proj, fc1, fc2};
  if (!h) return fail(SAMPT_ERR_ARG, "sampt_vit_set_gemm_workgroups_kind: null handle");
  for (int i = 0; i < 4; ++i)
    if (v[i] < 0 || v[i] > 32) return fail(SAMPT_ERR_ARG, "sampt_vit_set_gemm_workgroups_kind: counts must be 0 .. 32");
  for (int i = 0; i < 4; ++i) h->e.gemm_wgs_kind[i] = v[i];
  return SAMPT_OK;
}

int sampt_gemm_set_schedule(int sched) {
  if (sched < 0 || sched > 1) return fail(SAMPT_ERR_ARG, "sampt_gemm_set_schedule: 0 or 1");
  sampt::g_p8_sched = sched;
  return SAMPT_OK;
}

int sampt_gemm_set_thin_min_wgs(int n) {
  if (n < 1 || n > 4096) return fail(SAMPT_ERR_ARG, "sampt_gemm_set_thin_min_wgs: 1 .. 4096");
  sampt::g_thin_min_wgs = n;
  return SAMPT_OK;
}

int sampt_stream_create_cu_range(int cu_lo, int cu_hi, sampt_stream_t* out) {
  if (!out || cu_lo < 0 || cu_hi > 32 || cu_lo >= cu_hi) return fail(SAMPT_ERR_ARG, "sampt_stream_create_cu_range: need 0 <= cu_lo < cu_hi <= 32");
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return fail(SAMPT_ERR_HIP, "sampt_stream_create_cu_range: no device");
  const int ncu = p.multiProcessorCount, nxcd = 8;
  if (ncu != nxcd * 32) return fail(SAMPT_ERR_UNSUPPORTED, "sampt_stream_create_cu_range: written for 8 XCDs x 32 CUs");
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < ncu; ++i)              // bit i = CU i / 8 of XCD i % 8
    if (i / nxcd >= cu_lo && i / nxcd < cu_hi) mask[i / 32] |= 1u << (i % 32);
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
  if (e != hipSuccess) { sampt::set_error("hipExtStreamCreateWithCUMask", e); return SAMPT_ERR_HIP; }
  *out = (sampt_stream_t)s;
  return SAMPT_OK;
}

int sampt_stream_destroy(sampt_stream_t stream) {
  if (!stream) return SAMPT_OK;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? SAMPT_OK : SAMPT_ERR_HIP;
}

int sampt_pips_set_mixer(int fused, int workgroups) {
  if (fused < 0 || fused > 2 || workgroups < 1 || workgroups > 1024)
    return fail(SAMPT_ERR_ARG, "sampt_pips_set_mixer: fused 0 / 1 / 2, workgroups 1 .. 1024");
  sampt::g_pips_mixer_fused = fused ? 1 : 0, sampt::g_pips_mixer_x3 = fused == 2, sampt::g_pips_mixer_wgs = workgroups;
  const char* d = getenv("SAMPT_PIPS_MIXER_DIAG");      // measurement builds only: see pips_mixer.hip g_pips_mixer_diag
  sampt::g_pips_mixer_diag = d ? atoi(d) : 0;
  return SAMPT_OK;
}

int sampt_conv_set_halo(int on) {
  // 3: halo / stem kernels without the fused InstanceNorm statistics; 4: statistics from the halo kernel only; 5: from the stem only
  sampt::g_conv_in_stats = on == 3 ? 0 : (on == 4 ? 1 : (on == 5 ? 2 : 3));
  sampt::g_halo_dbg = on >= 6 && on <= 8 ? on : 0;
  sampt::g_conv_halo = on < 0 ? 0 : (on > 2 ? 1 : on);
  return SAMPT_OK;
}

int sampt_gemm_set_trim(int on) {
  sampt::g_p8_trim = on ? 1 : 0;
  return SAMPT_OK;
}

int sampt_gemm_set_wres(int on) {
  sampt::g_gemm_x3_wres = on ? 1 : 0;
  sampt::g_gemm_x3_epi = on == 2 ? 0 : 1;      // 2: the weights-resident kernel without the fused LayerNorm / dot-product tails
  return SAMPT_OK;
}

int sampt_gemm_set_stagger(int groups) {
  if (groups < 0 || groups > 8) return fail(SAMPT_ERR_ARG, "sampt_gemm_set_stagger: 0 .. 8 phase groups");
  sampt::g_p8_stagger = groups;
  return SAMPT_OK;
}

int sampt_vit_calibrate(sampt_vit_t h, float* colmeans_dev, int ld) {
  if (!h || (colmeans_dev && ld < h->e.c.mlp_ratio * h->e.c.D))
    return fail(SAMPT_ERR_ARG, "sampt_vit_calibrate: ld must be at least mlp_ratio * embed_dim");
  h->e.calib = colmeans_dev, h->e.calib_ld = colmeans_dev ? ld : 0;
  return SAMPT_OK;
}

int sampt_vit_profile_begin(sampt_vit_t h) {
  if (!h) return SAMPT_ERR_ARG;
  h->e.profiling = true;
  return SAMPT_OK;
}
int sampt_vit_profile_end(sampt_vit_t h, double* flop, double* ms, int* launches) {
  if (!h || !flop || !ms || !launches) return SAMPT_ERR_ARG;
  return h->e.profile_end(flop, ms, launches);
}

int sampt_vit_encode_workspace_bytes(sampt_vit_t h, int B, size_t* bytes) {
  if (!h || !bytes || B <= 0) return SAMPT_ERR_ARG;
  Arena a(nullptr, 0);
  int rc = h->e.encode(nullptr, 1, B, h->e.c.img, h->e.c.img, nullptr, nullptr, a, nullptr);
  // + the compact live-row stream of sampt_vit_encode_live (at most one more copy of the residual stream)
  *bytes = a.peak + 512 + (size_t)B * h->e.c.grid * h->e.c.grid * h->e.c.D * sizeof(float);
  return rc;
}

int sampt_vit_live_rows(sampt_vit_t h, int H, int W, int* live_rows, size_t* cache_bytes) {
  if (!h || !live_rows || !cache_bytes) return SAMPT_ERR_ARG;
  const int g = h->e.c.grid, lh = h->e.live_rows(H, W);
  *live_rows = lh;
  *cache_bytes = (size_t)(g - lh) * g * h->e.c.D * sizeof(float);
  return SAMPT_OK;
}

int sampt_vit_encode_live(sampt_vit_t h, const uint8_t* frames, int chw, int B, int H, int W, float* features,
                          float* interm_out, float* dead_cache, int build, void* ws, size_t ws_bytes,
                          sampt_stream_t stream) {
  if (!h || !frames || !dead_cache || !ws || B <= 0 || (!build && !features))
    return fail(SAMPT_ERR_ARG, "sampt_vit_encode_live: bad arguments");
  if (H > h->e.c.img || W > h->e.c.img || (H != h->e.c.img && W != h->e.c.img))
    return fail(SAMPT_ERR_UNSUPPORTED, "sampt_vit_encode_live: the frame's longest side must equal img_size");
  if (h->e.live_rows(H, W) >= h->e.c.grid)
    return fail(SAMPT_ERR_UNSUPPORTED,
                "sampt_vit_encode_live: this frame geometry has no frame-independent token rows (sampt_vit_live_rows)");
  Arena a(ws, ws_bytes);
  return h->e.encode(frames, chw, build ? 1 : B, H, W, features, interm_out, a, (hipStream_t)stream, dead_cache,
                     build ? 1 : 2);
}

int sampt_vit_encode(sampt_vit_t h, const uint8_t* frames, int chw, int B, int H, int W, float* features,
                     float* interm_out, void* ws, size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !frames || !features || !ws || B <= 0) return fail(SAMPT_ERR_ARG, "sampt_vit_encode: bad arguments");
  if (H > h->e.c.img || W > h->e.c.img || (H != h->e.c.img && W != h->e.c.img))
    return fail(SAMPT_ERR_UNSUPPORTED,
                "sampt_vit_encode: the frame's longest side must equal img_size (resize before SamPt, as the reference "
                "pipelines do)");
  Arena a(ws, ws_bytes);
  return h->e.encode(frames, chw, B, H, W, features, interm_out, a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------- PIPS++
int sampt_pips2_create(const char* const* names, const void* const* ptrs, int n, int stride, sampt_pips2_t* out) {
  if (!names || !ptrs || !out || stride <= 0) return fail(SAMPT_ERR_ARG, "sampt_pips2_create: bad arguments");
  return create_handle<sampt_pips2>("sampt_pips2_create", out,
                                    [&](sampt_pips2& h) { return h.e.init(make_map(names, ptrs, n), stride); });
}
void sampt_pips2_destroy(sampt_pips2_t h) { delete h; }

int sampt_pips2_fnet_workspace_bytes(sampt_pips2_t h, int nf, int H, int W, size_t* bytes) {
  if (!h || !bytes) return SAMPT_ERR_ARG;
  Arena a(nullptr, 0);
  float* out[4] = {nullptr, nullptr, nullptr, nullptr};
  int rc = h->e.enc.fnet(nullptr, nf, H, W, out, a, nullptr);
  *bytes = a.peak + 256;
  return rc;
}

int sampt_pips2_fnet_f32(sampt_pips2_t h, const void* frames, int frames_are_f32, int nf, int H, int W,
                         float* const pyr[4], void* ws, size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !frames || !pyr || !ws || H < 16 * h->e.stride || W < 16 * h->e.stride)
    return fail(SAMPT_ERR_ARG, "sampt_pips2_fnet_f32: bad arguments (H, W must be at least 16*stride)");
  Arena a(ws, ws_bytes);
  h->e.enc.frames_f32 = frames_are_f32 ? 1 : 0;
  return h->e.enc.fnet((const uint8_t*)frames, nf, H, W, pyr, a, (hipStream_t)stream);
}

int sampt_pips2_update_workspace_bytes(sampt_pips2_t h, int n, int S, size_t* bytes) {
  if (!h || !bytes || n <= 0 || S <= 0) return SAMPT_ERR_ARG;
  Arena a(nullptr, 0);
  PyramidLevels p = {};
  float* feats[3] = {nullptr, nullptr, nullptr};
  int rc = h->e.update(p, nullptr, n, S, nullptr, 0, feats, 1, nullptr, a, nullptr);
  *bytes = a.peak + 256;
  return rc;
}

static PyramidLevels make_pyr(const float* const pyr[4], int H0, int W0);

int sampt_pips2_update_f32(sampt_pips2_t h, const float* const pyr[4], int H0, int W0, const int32_t* frame_idx, int n,
                           int S, const float* trajs0, int have_feat_init, float* const feats[3], int iters,
                           float* trajs_out, void* ws, size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !pyr || !frame_idx || !trajs0 || !feats || !feats[0] || !feats[1] || !feats[2] || !trajs_out || !ws ||
      n <= 0 || S <= 0 || iters <= 0)
    return fail(SAMPT_ERR_ARG, "sampt_pips2_update_f32: bad arguments");
  Arena a(ws, ws_bytes);
  return h->e.update(make_pyr(pyr, H0, W0), (const int*)frame_idx, n, S, trajs0, have_feat_init, feats, iters, trajs_out,
                     a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------- decoder
int sampt_dec_create(const char* const* names, const void* const* ptrs, int n, int grid, int img_size, int max_frames,
                     int vit_dim, sampt_dec_t* out) {
  if (!names || !ptrs || !out || max_frames <= 0 || vit_dim < 0)
    return fail(SAMPT_ERR_ARG, "sampt_dec_create: bad arguments");
  DecConfig c;
  c.grid = grid, c.img = img_size, c.vit_dim = vit_dim;
  return create_handle<sampt_dec>("sampt_dec_create", out, [&](sampt_dec& h) {
    h.e.max_frames = max_frames;
    return h.e.init(make_map(names, ptrs, n), c);
  });
}
void sampt_dec_destroy(sampt_dec_t h) { delete h; }

int sampt_dec_workspace_bytes_k(sampt_dec_t h, int frames, int k, int oh, int ow, size_t* bytes) {
  if (!h || !bytes) return fail(SAMPT_ERR_ARG, "sampt_dec_workspace_bytes_k: null handle / output");
  if (frames <= 0 || frames > h->e.max_frames)
    return fail(SAMPT_ERR_ARG, "sampt_dec_workspace_bytes_k: frames " + std::to_string(frames) + " outside 1.." +
                                   std::to_string(h->e.max_frames) + " (max_frames of sampt_dec_create)");
  if (k < 0 || k > SAMPT_DEC_MAX_POINTS)
    return fail(SAMPT_ERR_ARG, "sampt_dec_workspace_bytes_k: k = " + std::to_string(k) + " prompt points outside 0.." +
                                   std::to_string(SAMPT_DEC_MAX_POINTS) + " (SAMPT_DEC_MAX_POINTS)");
  Arena a(nullptr, 0);
  float dummy = 0.f;
  int rc = h->e.track_decode(frames, &dummy, h->e.is_hq() ? &dummy : nullptr, &dummy, nullptr, k, nullptr, nullptr, k, 0, 1, 0.f, oh, ow,
                             oh, ow, nullptr, nullptr, a, nullptr);
  *bytes = a.peak + 256;
  return rc;
}

int sampt_dec_workspace_bytes(sampt_dec_t h, int frames, int oh, int ow, size_t* bytes) {
  return sampt_dec_workspace_bytes_k(h, frames, 120, oh, ow, bytes);
}

int sampt_dec_hq_workspace_bytes(sampt_dec_t h, int frames, size_t* bytes) {
  if (!h || !bytes || frames <= 0 || frames > h->e.max_frames) return SAMPT_ERR_ARG;
  if (!h->e.is_hq()) return fail(SAMPT_ERR_UNSUPPORTED, "sampt_dec_hq_workspace_bytes: not an HQ-SAM decoder handle");
  Arena a(nullptr, 0);
  int rc = h->e.hq_features(frames, nullptr, nullptr, nullptr, a, nullptr);
  *bytes = a.peak + 256;
  return rc;
}

int sampt_dec_hq_features(sampt_dec_t h, int frames, const float* features, const float* interm, float* hq_out, void* ws,
                          size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !features || !interm || !hq_out || !ws || frames <= 0 || frames > h->e.max_frames)
    return fail(SAMPT_ERR_ARG, "sampt_dec_hq_features: bad arguments");
  if (!h->e.is_hq()) return fail(SAMPT_ERR_UNSUPPORTED, "sampt_dec_hq_features: not an HQ-SAM decoder handle");
  Arena a(ws, ws_bytes);
  return h->e.hq_features(frames, features, interm, hq_out, a, (hipStream_t)stream);
}

int sampt_sam_decode(sampt_dec_t h, const float* features, const float* hq_features, const float* pts,
                     const int32_t* labels, int k, const float* box, const float* mask_in, int in_h, int in_w, int oh, int ow, float* logits_out,
                     float* iou_out, float* low_out, void* ws, size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !features || !logits_out || !iou_out || !low_out || !ws || k < 0 || (k > 0 && (!pts || !labels)))
    return fail(SAMPT_ERR_ARG, "sampt_sam_decode: bad arguments");
  if (h->e.is_hq() != (hq_features != nullptr))
    return fail(SAMPT_ERR_ARG, "sampt_sam_decode: hq_features must be given for HQ-SAM handles and only for them");
  Arena a(ws, ws_bytes);
  return h->e.decode(1, features, hq_features, pts, labels, k, nullptr, k > 0 ? k : 1, box, mask_in, in_h, in_w, oh, ow, logits_out, iou_out,
                     low_out, nullptr, a, (hipStream_t)stream);
}

int sampt_sam_decode_multimask(sampt_dec_t h, const float* features, const float* pts, const int32_t* labels, int k,
                               const float* box, const float* mask_in, int in_h, int in_w, int oh, int ow,
                               float* logits_out, float* iou_out, float* low_out, void* ws, size_t ws_bytes,
                               sampt_stream_t stream) {
  if (!h || !features || !logits_out || !iou_out || !low_out || !ws || k < 0 || (k > 0 && (!pts || !labels)))
    return fail(SAMPT_ERR_ARG, "sampt_sam_decode_multimask: bad arguments");
  if (h->e.is_hq()) return fail(SAMPT_ERR_UNSUPPORTED, "sampt_sam_decode_multimask: SAM decoder handles only");
  Arena a(ws, ws_bytes);
  h->e.multimask = true;
  int rc = h->e.decode(1, features, nullptr, pts, labels, k, nullptr, k > 0 ? k : 1, box, mask_in, in_h, in_w, oh, ow,
                       logits_out, iou_out, low_out, nullptr, a, (hipStream_t)stream);
  h->e.multimask = false;
  return rc;
}

int sampt_sam_track_decode(sampt_dec_t h, int frames, const float* features, const float* hq_features,
                           const float* pts, const int32_t* labels, int k, const int32_t* k_item,
                           const int32_t* npos_item, int ld_pts, int n_pos_first, int refine_iters, float iou_thr,
                           int in_h, int in_w, int oh, int ow, float* final_logits, float* score_out, void* ws,
                           size_t ws_bytes, sampt_stream_t stream) {
  if (!h || !features || !pts || !labels || !final_logits || !score_out || !ws || k <= 0 || n_pos_first > k ||
      ld_pts < k || frames <= 0 || frames > h->e.max_frames)
    return fail(SAMPT_ERR_ARG, "sampt_sam_track_decode: bad arguments");
  if (h->e.is_hq() != (hq_features != nullptr))
    return fail(SAMPT_ERR_ARG, "sampt_sam_track_decode: hq_features must be given for HQ-SAM handles and only for them");
  Arena a(ws, ws_bytes);
  if (npos_item && n_pos_first < 0)
    return fail(SAMPT_ERR_ARG, "sampt_sam_track_decode: npos_item needs the two-pass mode (n_pos_first >= 0)");
  return h->e.track_decode(frames, features, hq_features, pts, labels, k, k_item, npos_item, ld_pts, n_pos_first,
                           refine_iters, iou_thr, in_h, in_w, oh, ow, final_logits, score_out, a, (hipStream_t)stream);
}

int sampt_sam_track_decode_graph(sampt_dec_t h, int frames, const float* features, const float* hq_features,
                                 const float* pts, const int32_t* labels, int k, const int32_t* k_item,
                                 const int32_t* npos_item, int ld_pts, int n_pos_first, int refine_iters, float iou_thr,
                                 int in_h, int in_w, int oh, int ow, float* final_logits, float* score_out, void* ws,
                                 size_t ws_bytes, sampt_stream_t stream) {
  if (!h) return fail(SAMPT_ERR_ARG, "sampt_sam_track_decode_graph: null handle");
  hipStream_t s = (hipStream_t)stream;
  if (s == nullptr)    // the legacy default stream cannot be captured
    return sampt_sam_track_decode(h, frames, features, hq_features, pts, labels, k, k_item, npos_item, ld_pts, n_pos_first,
                                  refine_iters, iou_thr, in_h, in_w, oh, ow, final_logits, score_out, ws, ws_bytes, stream);
  DecGraphKey key;
  memset(&key, 0, sizeof(key));   // padding bytes are part of the hash / comparison
  key.features = features, key.hq = hq_features, key.pts = pts, key.labels = labels, key.k_item = k_item;
  key.npos_item = npos_item, key.logits = final_logits, key.score = score_out, key.ws = ws, key.ws_bytes = ws_bytes;
  key.frames = frames, key.k = k, key.ld_pts = ld_pts, key.n_pos_first = n_pos_first, key.refine = refine_iters;
  key.in_h = in_h, key.in_w = in_w, key.oh = oh, key.ow = ow, key.iou_thr = iou_thr;
  try {
    DecGraphEntry& ent = h->graphs[key];
    ent.last_use = ++h->graph_clock;
    while (h->graphs.size() > 64) {   // bound the cache: drop the least recently used signature, captured or only seen
      auto victim = h->graphs.end();
      for (auto it = h->graphs.begin(); it != h->graphs.end(); ++it)
        if (&it->second != &ent && (victim == h->graphs.end() || it->second.last_use < victim->second.last_use)) victim = it;
      if (victim == h->graphs.end()) break;
      if (victim->second.exec) (void)hipGraphExecDestroy(victim->second.exec);
      h->graphs.erase(victim);       // (references to other elements of an unordered_map stay valid)
    }
    if (ent.exec) {
      if (hipGraphLaunch(ent.exec, s) != hipSuccess) return fail(SAMPT_ERR_HIP, "sampt_sam_track_decode_graph: hipGraphLaunch failed");
      ++h->graph_launches;
      return SAMPT_OK;
    }
    if (ent.seen++ == 0)   // first sight of a signature: plain launches (validates the arguments, sets lazy attributes)
      return sampt_sam_track_decode(h, frames, features, hq_features, pts, labels, k, k_item, npos_item, ld_pts,
                                    n_pos_first, refine_iters, iou_thr, in_h, in_w, oh, ow, final_logits, score_out, ws,
                                    ws_bytes, stream);
    DecGraphEntry& e2 = h->graphs[key];
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess)
      return fail(SAMPT_ERR_HIP, "sampt_sam_track_decode_graph: hipStreamBeginCapture failed");
    int rc = sampt_sam_track_decode(h, frames, features, hq_features, pts, labels, k, k_item, npos_item, ld_pts, n_pos_first,
                                    refine_iters, iou_thr, in_h, in_w, oh, ow, final_logits, score_out, ws, ws_bytes, stream);
    hipGraph_t g = nullptr;
    hipError_t ce = hipStreamEndCapture(s, &g);
    if (rc != SAMPT_OK || ce != hipSuccess || !g) {
      if (g) (void)hipGraphDestroy(g);
      return rc != SAMPT_OK ? rc : fail(SAMPT_ERR_HIP, "sampt_sam_track_decode_graph: stream capture failed");
    }
    hipGraphExec_t exec = nullptr;
    hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess || !exec) return fail(SAMPT_ERR_HIP, "sampt_sam_track_decode_graph: hipGraphInstantiate failed");
    e2.exec = exec;
    ++h->graph_captures;
    if (hipGraphLaunch(exec, s) != hipSuccess) return fail(SAMPT_ERR_HIP, "sampt_sam_track_decode_graph: hipGraphLaunch failed");
    ++h->graph_launches;
    return SAMPT_OK;
  } catch (...) {
    return fail(SAMPT_ERR_ARG, "sampt_sam_track_decode_graph: C++ exception");
  }
}

int sampt_dec_graph_stats(sampt_dec_t h, long* cached, long* captures, long* launches) {
  if (!h || !cached || !captures || !launches) return SAMPT_ERR_ARG;
  *cached = (long)h->graphs.size(), *captures = h->graph_captures, *launches = h->graph_launches;
  return SAMPT_OK;
}

int sampt_postprocess_masks(const float* low, int L, int img, int in_h, int in_w, float* out, int oh, int ow,
                            sampt_stream_t stream) {
  return sam_postprocess(low, L, img, in_h, in_w, out, oh, ow, (hipStream_t)stream);
}

size_t sampt_bbox_workspace_bytes(int h, int w) { return bbox_partial_ints(h, w) * sizeof(int); }

int sampt_bbox_from_logits(const float* logits, int h, int w, int32_t* bbox_state, void* ws, size_t ws_bytes,
                           sampt_stream_t stream) {
  if (!logits || !bbox_state || !ws || ws_bytes < sampt_bbox_workspace_bytes(h, w)) return SAMPT_ERR_WORKSPACE;
  return bbox_from_logits_state(logits, h, w, (int*)bbox_state, (int*)ws, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------- kernel-level
int sampt_gemm(int dtype, const void* A, const void* W, const float* bias, const float* res, void* C, int M, int N, int K,
               int act, float alpha, sampt_stream_t stream) {
  GemmP p;
  p.A = A, p.W = W, p.bias = bias, p.res = res, p.C = C;
  p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldc = N, p.ldr = N, p.act = act, p.alpha = alpha;
  p.out_f16 = dtype == 2;
  return dtype == 0 ? gemm_f32(p, (hipStream_t)stream) : gemm_f16(p, (hipStream_t)stream);
}

int sampt_gemm_ex(int dtype, const void* A, const void* W, const float* bias, const float* res, void* C, int M, int N, int K,
                  int act, float alpha, const int32_t* rowmap, const int32_t* a_rowmap, int res_mod, int ldr,
                  sampt_stream_t stream) {
  GemmP p;
  p.A = A, p.W = W, p.bias = bias, p.res = res, p.C = C, p.rowmap = rowmap, p.a_rowmap = a_rowmap;
  p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldc = N, p.ldr = ldr > 0 ? ldr : N, p.act = act, p.alpha = alpha;
  p.res_mod = res_mod;
  p.out_f16 = dtype == 2;
  if (dtype == 3 || dtype == 4) {   // x3 rows in (A, W: 2K halves per row), f32 or x3 rows out
    p.x3 = 1, p.K = 2 * K, p.lda = 2 * K, p.ldw = 2 * K;
    if (dtype == 4) p.out_f16 = 2, p.ldc = 2 * N;
  }
  return dtype == 0 ? gemm_f32(p, (hipStream_t)stream) : gemm_f16(p, (hipStream_t)stream);
}

int sampt_kmedoids_rowsums_f64(const float* xy, int n, double* out, sampt_stream_t stream) {
  int rc = kmedoids_rowsums(xy, n, out, (hipStream_t)stream);
  return rc == SAMPT_ERR_ARG ? fail(rc, "sampt_kmedoids_rowsums_f64: bad arguments (1 <= n <= 2048)") : rc;
}

int sampt_kmedoids_alternate(const float* xy, int n, int K, int32_t* medoids, int max_iter, int32_t* iters_out,
                             sampt_stream_t stream) {
  int rc = kmedoids_alternate(xy, n, K, (int*)medoids, max_iter, (int*)iters_out, (hipStream_t)stream);
  return rc == SAMPT_ERR_ARG ? fail(rc, "sampt_kmedoids_alternate: bad arguments (1 <= K <= min(n, 64), n <= 2048)") : rc;
}

int sampt_qp_corners_workspace_bytes(int H, int W, size_t* bytes) {
  if (!bytes || H < 3 || W < 3) return fail(SAMPT_ERR_ARG, "sampt_qp_corners_workspace_bytes: bad arguments");
  *bytes = qp_corners_workspace_bytes(H, W);
  return SAMPT_OK;
}

int sampt_qp_erode_u8(const uint8_t* mask, int H, int W, int k, uint8_t* tmp, uint8_t* out, sampt_stream_t stream) {
  int rc = qp_erode(mask, H, W, k, tmp, out, (hipStream_t)stream);
  return rc == SAMPT_ERR_ARG ? fail(rc, "sampt_qp_erode_u8: bad arguments") : rc;
}

int sampt_qp_shi_tomasi(const uint8_t* image, const uint8_t* mask, int H, int W, int n_points, float quality_level, float* out_xy,
                        int32_t* out_info, void* ws, size_t ws_bytes, sampt_stream_t stream) {
  int rc = qp_corners(image, mask, H, W, n_points, quality_level, out_xy, (int*)out_info, ws, ws_bytes, (hipStream_t)stream);
  return rc == SAMPT_ERR_ARG ? fail(rc, "sampt_qp_shi_tomasi: bad arguments (H, W >= 3, 1 <= n_points <= 64, workspace of "
                                        "sampt_qp_corners_workspace_bytes)") : rc;
}

int sampt_split_rows_x3(const float* x, void* y, int M, int K, sampt_stream_t stream) {
  if (!x || !y) return fail(SAMPT_ERR_ARG, "sampt_split_rows_x3: bad arguments");
  return split_rows_x3(x, (half_t*)y, M, K, (hipStream_t)stream);
}

int sampt_conv2d_nhwc(int dtype, const void* x, const void* w, const float* bias, float* y, int n, int H, int W, int Cin,
                      int Cout, int KH, int KW, int stride, int pad, sampt_stream_t stream) {
  GemmP p;
  p.OH = (H + 2 * pad - KH) / stride + 1, p.OW = (W + 2 * pad - KW) / stride + 1;
  p.A = x, p.W = w, p.bias = bias, p.C = y;
  p.M = n * p.OH * p.OW, p.N = Cout, p.K = KH * KW * Cin, p.ldw = p.K, p.ldc = Cout;
  p.conv = 1, p.cH = H, p.cW = W, p.cC = Cin, p.KH = KH, p.KW = KW, p.cstride = stride, p.cpad = pad;
  if (dtype == 3 || dtype == 4) {
    p.W_lo = (const half_t*)w + (size_t)Cout * p.K;
    p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
    if (dtype == 4) p.A_lo = (const half_t*)x + (size_t)n * H * W * Cin;   // activations pre-split: [2][n][H][W][Cin] halves
    return conv_f16x3(p, (hipStream_t)stream);
  }
  return dtype == 0 ? gemm_f32(p, (hipStream_t)stream) : gemm_f16(p, (hipStream_t)stream);
}

int sampt_gemm_x3_rows(const float* A, const void* w_hl, const float* bias, const float* res, int res_mod, float* C, int M, int N,
                       int K, int act, int shuf_g, sampt_stream_t stream) {
  return sampt_gemm_x3_rows_epi(A, w_hl, bias, res, res_mod, C, M, N, K, act, shuf_g, 0, nullptr, nullptr, 0.f, 0, stream);
}

int sampt_gemm_x3_rows_epi(const float* A, const void* w_hl, const float* bias, const float* res, int res_mod, float* C, int M, int N,
                           int K, int act, int shuf_g, int epi, const float* epi_a, const float* epi_b, float epi_eps, int epi_ld,
                           sampt_stream_t stream) {
  GemmP p;
  p.A = A, p.W = w_hl, p.W_lo = (const half_t*)w_hl + (size_t)N * K, p.bias = bias, p.res = res, p.res_mod = res_mod, p.C = C;
  p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT), p.act = act;
  p.M = M, p.N = N, p.K = K, p.ldw = K, p.ldc = epi == 2 ? 1 : (shuf_g ? N / 4 : N), p.ldr = shuf_g ? N / 4 : N;
  p.conv = 1, p.cH = M, p.cW = 1, p.cC = K, p.KH = 1, p.KW = 1, p.cstride = 1, p.cpad = 0, p.OH = M, p.OW = 1;
  p.shuf_g = shuf_g, p.shuf_n = shuf_g ? N / 4 : 0;
  p.epi = epi, p.epi_a = epi_a, p.epi_b = epi_b, p.epi_eps = epi_eps, p.epi_ld = epi_ld;
  if (epi) {       // the fused tails exist in the weights-resident kernel only: refuse rather than compute something else
    if (!sampt::g_gemm_x3_wres || !(epi == 3 ? gemm_x3_wres_ln_eligible(p) : gemm_x3_wres_eligible(p))) return fail(SAMPT_ERR_UNSUPPORTED, "sampt_gemm_x3_rows_epi: shape without a fused tail");
    return gemm_x3_wres(p, (hipStream_t)stream);
  }
  return conv_f16x3(p, (hipStream_t)stream);
}

int sampt_sam_mask_dot(const float* up, const float* hyper, int ld_hyper, float* low, int frames, int npix, int C, sampt_stream_t stream) {
  return sam_mask_dot(up, hyper, ld_hyper, nullptr, nullptr, 0, low, frames, npix, C, (hipStream_t)stream);
}

int sampt_move_rows(const void* src, void* dst, const int* idx, int rows, size_t row_bytes, int n_objects, int n_frames, int scatter,
                    sampt_stream_t stream) {
  return move_rows(src, dst, idx, rows, (long)row_bytes, n_objects, n_frames, scatter, (hipStream_t)stream);
}

int sampt_fill_f32(float* dst, size_t n, float value, sampt_stream_t stream) { return fill_f32(dst, (long)n, value, (hipStream_t)stream); }

int sampt_conv_stem7x7(const float* x_nhwc4, const float* w, const float* bias, float* y, int n, int H, int W, float eps,
                       float* mean_rstd, void* ws, size_t ws_bytes, sampt_stream_t stream) {
  const int chunks = conv_stem_tiles(H, W);
  if (mean_rstd && ws_bytes < (size_t)n * chunks * 64 * 2 * sizeof(double)) return SAMPT_ERR_WORKSPACE;
  SAMPT_TRY(conv_stem7x7_x3(x_nhwc4, w, bias, y, n, H, W, mean_rstd ? (double*)ws : nullptr, (hipStream_t)stream));
  if (!mean_rstd) return SAMPT_OK;
  const long hw = (long)((H + 6 - 7) / 2 + 1) * ((W + 6 - 7) / 2 + 1);
  return instnorm_finalize((const double*)ws, n, chunks, hw, 64, eps, mean_rstd, (hipStream_t)stream);
}

int sampt_conv3x3_planes_instnorm_stats(const void* x_hl, const void* w_hl, const float* bias, float* y, int n, int H, int W, int Cin,
                                        int Cout, float eps, float* mean_rstd, void* ws, size_t ws_bytes, sampt_stream_t stream) {
  GemmP p;
  p.OH = H, p.OW = W;
  p.A = x_hl, p.A_lo = (const half_t*)x_hl + (size_t)n * H * W * Cin, p.W = w_hl, p.W_lo = (const half_t*)w_hl + (size_t)Cout * 9 * Cin;
  p.bias = bias, p.C = y, p.alpha = 1.0f / (float)(1 << F16X3_WSHIFT);
  p.M = n * H * W, p.N = Cout, p.K = 9 * Cin, p.ldw = p.K, p.ldc = Cout;
  p.conv = 1, p.cH = H, p.cW = W, p.cC = Cin, p.KH = 3, p.KW = 3, p.cstride = 1, p.cpad = 1;
  if (!conv3x3_halo_eligible(p)) return fail(SAMPT_ERR_UNSUPPORTED, "sampt_conv3x3_planes_instnorm_stats: Cin % 32, Cout % 4");
  const int chunks = conv3x3_halo_tiles(p);
  if (ws_bytes < (size_t)n * chunks * Cout * 2 * sizeof(double)) return SAMPT_ERR_WORKSPACE;
  p.in_part = (double*)ws;
  SAMPT_TRY(conv3x3_halo_x3(p, (hipStream_t)stream));
  return instnorm_finalize(p.in_part, n, chunks, (long)H * W, Cout, eps, mean_rstd, (hipStream_t)stream);
}

size_t sampt_instance_norm_workspace_bytes(int n, int hw, int C) {
  return instnorm_partial_doubles(n, hw, C) * sizeof(double) + (size_t)n * C * 2 * sizeof(float) + 512;
}

int sampt_instance_norm_nhwc(float* x, int n, int hw, int C, float eps, int relu, const float* skip, void* ws,
                             size_t ws_bytes, sampt_stream_t stream) {
  if (ws_bytes < sampt_instance_norm_workspace_bytes(n, hw, C)) return SAMPT_ERR_WORKSPACE;
  Arena a(ws, ws_bytes);
  double* part = (double*)a.get(instnorm_partial_doubles(n, hw, C) * sizeof(double));
  float* mr = a.f32((size_t)n * C * 2);
  SAMPT_TRY(instnorm_stats(x, n, hw, C, eps, part, mr, (hipStream_t)stream));
  return instnorm_apply(x, mr, skip, x, n, hw, C, relu, (hipStream_t)stream);
}

int sampt_layernorm(const float* x, const float* w, const float* b, void* y, int M, int D, float eps, int out_f16,
                    int act, sampt_stream_t stream) {
  return layernorm_rows(x, w, b, y, M, D, eps, nullptr, out_f16, act, (hipStream_t)stream);
}

int sampt_resize_bilinear_nhwc(const float* src, int n, int sh, int sw, int C, float* dst, int dh, int dw, int dstC,
                               int c_off, int align_corners, sampt_stream_t stream) {
  return resize_bilinear_nhwc(src, n, sh, sw, C, dst, dh, dw, dstC, c_off, align_corners, (hipStream_t)stream);
}

int sampt_avgpool2x2_nhwc(const float* src, int n, int h, int w, int C, float* dst, sampt_stream_t stream) {
  return avgpool2x2_nhwc(src, n, h, w, C, dst, (hipStream_t)stream);
}

int sampt_resize_logits(const float* src, int n, int sh, int sw, float* dst, int dh, int dw, sampt_stream_t stream) {
  if (!src || !dst || n <= 0) return SAMPT_ERR_ARG;
  return resize_logits(src, n, sh, sw, dst, dh, dw, (hipStream_t)stream);
}

int sampt_vos_index_masks(const float* logits, int M, int T, long hw, const int32_t* query_t, const uint8_t* gt_masks,
                          uint8_t* out, sampt_stream_t stream) {
  if (!logits || !out || !query_t || T <= 0 || hw <= 0) return SAMPT_ERR_ARG;
  return vos_index_masks(logits, M, T, hw, (const int*)query_t, gt_masks, out, (hipStream_t)stream);
}

int sampt_pil_resample_u8(const uint8_t* src, uint8_t* dst, long outer, int in_len, int out_len, int inner,
                          const int32_t* coef, const int32_t* bounds, int ksize, sampt_stream_t stream) {
  if (!src || !dst || !coef || !bounds || outer <= 0 || in_len <= 0 || out_len <= 0 || inner <= 0)
    return fail(SAMPT_ERR_ARG, "sampt_pil_resample_u8: bad arguments");
  return pil_resample_u8(src, dst, outer, in_len, out_len, inner, (const int*)coef, (const int*)bounds, ksize,
                         (hipStream_t)stream);
}

int sampt_vos_index_masks_resized(const float* logits, int M, int T, int h, int w, const int32_t* query_t,
                                  const uint8_t* gt_masks, int oh, int ow, uint8_t* out, sampt_stream_t stream) {
  if (!logits || !out || !query_t || T <= 0 || h <= 0 || w <= 0) return SAMPT_ERR_ARG;
  return vos_index_masks_resized(logits, M, T, h, w, (const int*)query_t, gt_masks, oh, ow, out, (hipStream_t)stream);
}

int sampt_index_masks(const float* logits, int M, long npix, uint8_t* out, sampt_stream_t stream) {
  if (!logits || !out || npix <= 0) return SAMPT_ERR_ARG;
  return index_masks(logits, M, npix, out, (hipStream_t)stream);
}

int sampt_corr_sample_f32(const float* const pyr[4], int H0, int W0, const int32_t* frame_idx, int S, int n,
                          const float* ffeats, const float* coords, float* out, sampt_stream_t stream) {
  return pips_corr_sample(make_pyr(pyr, H0, W0), frame_idx, S, n, 128, ffeats, coords, out, 196, 0, (hipStream_t)stream);
}

int sampt_pips_mix_mlp_f32(const float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                           float* part, int nseq, int slices, sampt_stream_t stream) {
  if (!x || !lnw || !lnb || !w1 || !b1 || !w2 || !part) return fail(SAMPT_ERR_ARG, "sampt_pips_mix_mlp_f32: null pointer");
  return pips_mix_mlp(x, lnw, lnb, w1, b1, w2, part, nseq, slices, (hipStream_t)stream);
}

int sampt_pips_mix_reduce_f32(const float* part, int slices, const float* bias, const float* res, int nseq, int mode,
                              const float* lnw, const float* lnb, const float* tw1, const float* tb1, const float* tw2,
                              const float* tb2, float* out, sampt_stream_t stream) {
  if (!lnw || !lnb || (mode != 0 && mode != 1) || (mode == 0 && (!tw1 || !tb1 || !tw2 || !tb2)))
    return fail(SAMPT_ERR_ARG, "sampt_pips_mix_reduce_f32: bad arguments");
  return pips_mix_reduce(part, slices, bias, res, nseq, mode, lnw, lnb, tw1, tb1, tw2, tb2, out, (hipStream_t)stream);
}

size_t sampt_pips_mix_xop_halves(int nseq) { return nseq > 0 ? pips_mix_xop_halves(nseq) : 0; }

int sampt_pips_mix_pre_f32(const float* part, int slices, const float* bias, const float* res, int nseq, const float* ln1w,
                           const float* ln1b, const float* tw1, const float* tb1, const float* tw2, const float* tb2,
                           const float* ln2w, const float* ln2b, float* xout, void* xop, sampt_stream_t stream) {
  if (!ln1w || !ln1b || !tw1 || !tb1 || !tw2 || !tb2 || !ln2w || !ln2b) return fail(SAMPT_ERR_ARG, "sampt_pips_mix_pre_f32: null pointer");
  return pips_mix_pre(part, slices, bias, res, nseq, ln1w, ln1b, tw1, tb1, tw2, tb2, ln2w, ln2b, xout, (half_t*)xop, (hipStream_t)stream);
}

int sampt_pips_mix_mlp_x3(const void* xop, const void* wstream, const float* b1, float* part, int nseq, int slices,
                          sampt_stream_t stream) {
  return pips_mix_mlp_x3((const half_t*)xop, (const half_t*)wstream, b1, part, nseq, slices, (hipStream_t)stream);
}

int sampt_vit_attention_f16(const void* qkv, const float* rel_h, const float* rel_w, void* out, int B, int S, int heads,
                            int hd, void* ws, size_t ws_bytes, sampt_stream_t stream) {
  (void)ws, (void)ws_bytes;  // the decomposed rel-pos bias is computed inside the kernel: no scratch needed any more
  return vit_flash_attention_f16((const half_t*)qkv, rel_h, rel_w, (half_t*)out, B, S, heads, hd, (hipStream_t)stream);
}

int sampt_vit_attention_x3(const void* qkv, const float* rel_h, const float* rel_w, void* out, int B, int S, int heads,
                           int hd, sampt_stream_t stream) {
  if (!qkv || !rel_h || !rel_w || !out) return fail(SAMPT_ERR_ARG, "sampt_vit_attention_x3: bad arguments");
  return vit_flash_attention_x3((const half_t*)qkv, rel_h, rel_w, (half_t*)out, B, S, heads, hd, (hipStream_t)stream);
}

int sampt_vit_window_attention(int precision, const void* qkv, const float* rel_h, const float* rel_w, void* out, int frames,
                               int S, int heads, int hd, const void* bias_row, int nwx, int nwy, int grid_h, int grid_w,
                               sampt_stream_t stream) {
  if (!qkv || !rel_h || !rel_w || !out || !bias_row || frames <= 0 || nwx <= 0 || nwy <= 0)
    return fail(SAMPT_ERR_ARG, "sampt_vit_window_attention: bad arguments");
  FlashPad fp;
  fp.bias_row = (const half_t*)bias_row, fp.nwx = nwx, fp.nwin = nwx * nwy, fp.gh = grid_h, fp.gw = grid_w;
  const int B = frames * nwx * nwy;
  if (precision == 1)
    return vit_flash_attention_f16((const half_t*)qkv, rel_h, rel_w, (half_t*)out, B, S, heads, hd, (hipStream_t)stream, fp);
  if (precision == 2)
    return vit_flash_attention_x3((const half_t*)qkv, rel_h, rel_w, (half_t*)out, B, S, heads, hd, (hipStream_t)stream, fp);
  return fail(SAMPT_ERR_ARG, "sampt_vit_window_attention: precision must be 1 (fp16) or 2 (x3 rows)");
}

int sampt_attention_f32(int kind, const float* q, const float* k, const float* v, float* out, int F, int Nq, int Nk,
                        int heads, int hd, const int32_t* nk_item, sampt_stream_t stream) {
  if (!q || !k || !v || !out) return fail(SAMPT_ERR_ARG, "sampt_attention_f32: bad arguments");
  if (kind == 0) return attn_rowblock(q, k, v, out, F, Nq, Nk, heads, hd, (const int*)nk_item, (hipStream_t)stream);
  if (kind == 1) return attn_fewkeys(q, k, v, out, F, Nq, Nk, heads, hd, (const int*)nk_item, (hipStream_t)stream);
  return SAMPT_ERR_UNSUPPORTED;
}

int sampt_attention_t2i_workspace_bytes(int F, int Nq, int Nk, size_t* bytes) {
  if (!bytes || F <= 0 || Nq <= 0 || Nk <= 0) return SAMPT_ERR_ARG;
  *bytes = attn_t2i_workspace_floats(F, Nq, Nk) * sizeof(float) + 16;
  return SAMPT_OK;
}

int sampt_attention_t2i_f32(const float* q, const float* k, const float* v, float* out, int F, int Nq, int Nk, void* ws,
                            size_t ws_bytes, sampt_stream_t stream) {
  if (!q || !k || !v || !out) return fail(SAMPT_ERR_ARG, "sampt_attention_t2i_f32: bad arguments");
  return attn_t2i(q, k, v, out, F, Nq, Nk, (float*)ws, ws_bytes / sizeof(float), (hipStream_t)stream);
}

int sampt_cotracker_attention_f32(const float* qkv, float* out, int nbatch, int L, int batch_stride_rows,
                                  int token_stride_rows, int heads, int hd, sampt_stream_t stream) {
  if (!qkv || !out) return fail(SAMPT_ERR_ARG, "sampt_cotracker_attention_f32: bad arguments");
  return cot_attention(qkv, out, nbatch, L, batch_stride_rows, token_stride_rows, heads, hd, (hipStream_t)stream);
}

}  // extern "C"
