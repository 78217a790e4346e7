"""GPU parity tests, module level: the HIP engines behind the two drop-in seams against the CPU oracle
(oracle/pips_ref.py pinned to the reference's own PIPS; oracle/sam_ref.py pinned to HF transformers.sam) on the same
seeded weights and inputs.  Tolerances are stated per test; trajectories must agree in index space
(round(traj) identical), masks within 1e-3 IoU."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.util import disc_queries, iou, max_abs, rel_err, synthetic_clip

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ PIPS
@pytest.fixture(scope="module")
def pips_sd():
    from sam_pt_amd.weights import init_pips_state_dict
    return init_pips_state_dict(72)


@pytest.fixture(scope="module")
def clip():
    return synthetic_clip(T=12, H=128, W=256, seed=72)


def test_fnet_vs_oracle(dev, pips_sd, clip):
    """fnet + pyramid (exact-f32 MFMA implicit-GEMM convs, fp64-accumulated instance norm) vs pips.py:254-287."""
    from oracle import pips_ref as O
    from sam_pt_amd.point_tracker import PipsPointTracker
    frames, _ = clip
    trk = PipsPointTracker(state_dict=pips_sd, fnet_chunk=2)
    pyr = trk.compute_pyramid(frames[:3].to(dev))
    ref = O.fnet(pips_sd, O.normalize_rgbs(frames[:3]), 4)
    refp = O.build_pyramid(ref)
    for l in range(4):
        got = pyr[l].permute(0, 3, 1, 2)
        assert got.shape == refp[l].shape
        assert rel_err(got, refp[l]) < 2e-5, f"level {l}"


def test_update_window_vs_oracle(dev, pips_sd, clip):
    """One 8-frame window (6 iterations: corr sampler, mixer, feature/coord update, vis head) vs Pips.forward."""
    from oracle import pips_ref as O
    from sam_pt_amd import _lib
    from sam_pt_amd.point_tracker import PipsPointTracker
    frames, centres = clip
    q = disc_queries(centres, n_pos=5, r=9.0)
    trk = PipsPointTracker(state_dict=pips_sd)
    pyr = trk.compute_pyramid(frames[:8].to(dev))
    fm = pyr[0].permute(0, 3, 1, 2).cpu()                      # feed the oracle OUR fmaps: isolates the window
    xys = q[:, 1:]
    preds, vlog, ffeat = O.pips_forward(pips_sd, xys, fm, None, iters=6)
    lib = _lib.load()
    nb = C.c_size_t()
    _lib.check(lib.sampt_pips_update_workspace_bytes(trk._h, 5, C.byref(nb)), "ws")
    ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    fi = torch.empty(5, 128, device=dev)
    xy0 = (xys / 4.0).contiguous().to(dev)
    _lib.check(lib.sampt_pips_sample_feat_f32(_lib.ptr(pyr[0]), 32, 64, None, _lib.ptr(xy0), 5, _lib.ptr(fi), _lib.stream_ptr()), "feat")
    assert max_abs(fi, ffeat) < 1e-5
    fidx = torch.arange(8, dtype=torch.int32, device=dev).repeat(5, 1).contiguous()   # per-point window frames [n][S]
    tr = torch.empty(8, 5, 2, device=dev)
    vi = torch.empty(8, 5, device=dev)
    xys_d = xys.contiguous().to(dev)
    _lib.check(lib.sampt_pips_update_f32(trk._h, _lib.ptr_array(pyr), 32, 64, _lib.ptr(fidx), 5, _lib.ptr(xys_d),
                                         _lib.ptr(fi), 6, _lib.ptr(tr), _lib.ptr(vi), _lib.ptr(ws), nb.value, _lib.stream_ptr()), "update")
    assert max_abs(tr, preds[-1]) < 2e-3, "trajectory (px)"
    assert max_abs(vi, torch.sigmoid(vlog)) < 1e-4


def test_tracker_vs_oracle(dev, pips_sd, clip):
    """PipsPointTracker.forward (both directions, linking) vs the oracle tracker == the reference's tracker."""
    from oracle import pips_ref as O
    from sam_pt_amd.point_tracker import PipsPointTracker
    frames, centres = clip
    q = torch.cat([disc_queries(centres, n_pos=4, r=9.0, t=0), disc_queries(centres, n_pos=2, r=6.0, t=5),
                   disc_queries(centres, n_pos=1, r=3.0, t=11)])[None]
    rgbs = frames[None]
    tr_ref, vi_ref = O.PipsTrackerRef(pips_sd).forward(rgbs, q)
    trk = PipsPointTracker(state_dict=pips_sd)
    out = trk.evaluate_batch(rgbs.to(dev), q.to(dev))
    tr, vi = out["trajectories_pred"], out["visibilities_pred"]
    assert tr.shape == (1, 12, 7, 2) and vi.shape == (1, 12, 7)
    assert (vi.bool() == vi_ref).all(), "visibilities differ"
    assert max_abs(tr, tr_ref) < 5e-3
    assert (tr.round() == tr_ref.round()).all(), "trajectories differ in index space"


# ------------------------------------------------------------------------------------------ SAM
def _sam(variant, precision, seed=72, max_batch=2):
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    return SamPredictor(SamHip(variant, precision=precision, seed=seed, max_batch=max_batch).cuda())


@pytest.mark.parametrize("precision,tol", [("f32", 3e-5), ("f16", 2e-2), ("f16x3", 3e-5)])
def test_vit_test_encoder_vs_oracle(dev, precision, tol):
    """Reduced geometry (2 blocks: 1 windowed with padding 16->18, 1 global), batch 2, non-square frame."""
    from oracle import sam_ref as R
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    frames, _ = synthetic_clip(T=2, H=144, W=256, seed=5)
    pred = _sam("vit_test", precision)
    feats = pred.encode_frames(frames.to(dev))                                    # (2, 256, 256)
    ref = R.image_encoder(sd, cfg, R.preprocess(cfg, frames.float()))             # (2,256,16,16)
    got = feats.view(2, 16, 16, 256).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < tol


@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("f16", 2e-3), ("f16x3", 2e-5)])     # measured 2.6e-6 / 7.4e-4 / see profiles/r4_*
def test_vit_b_encoder_vs_oracle(dev, precision, tol):
    from oracle import sam_ref as R
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_b"]
    sd = init_sam_state_dict(cfg, 72)
    frames, _ = synthetic_clip(T=1, H=576, W=1024, seed=9, disc_r=60)
    pred = _sam("vit_b", precision, max_batch=1)
    feats = pred.encode_frames(frames.to(dev))
    ref = R.image_encoder(sd, cfg, R.preprocess(cfg, frames.float()))
    got = feats.view(1, 64, 64, 256).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < tol


def test_vit_b_f16_static_bias_correction(dev, monkeypatch):
    """The fp16 mode's static bias correction (SamPredictor._select_bias_set: the token-mean part of the weight-rounding error
    folded into the biases, calibrated per frame geometry on a seeded noise frame — VERDICT r3 / r4 "rank-1 correction"): the
    embedding error against the fp32 oracle drops by about a third on a frame the calibration never saw (CPU emulation: rms
    6.0e-4 -> 3.9e-4); the result depends on the frame geometry only, not on what was encoded before (a second predictor that
    first encodes another geometry gives the same bits); the qkv bias row of padded window tokens is untouched."""
    from oracle import sam_ref as R
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_b"]
    sd = init_sam_state_dict(cfg, 72)
    frames, _ = synthetic_clip(T=1, H=576, W=1024, seed=9, disc_r=60)
    ref = R.image_encoder(sd, cfg, R.preprocess(cfg, frames.float()))
    rms = lambda e: float(((e - ref).double().pow(2).mean().sqrt()) / ref.double().pow(2).mean().sqrt())
    as_img = lambda f: f.view(1, 64, 64, 256).permute(0, 3, 1, 2).cpu()
    pred = _sam("vit_b", "f16", max_batch=1)
    assert pred.bias_correction
    e_corr = as_img(pred.encode_frames(frames.to(dev)))
    assert pred.stats["bias_calibrations"] == 1
    row16 = pred._wv["image_encoder.blocks.0.attn.qkv.bias.f16"].clone()
    assert torch.equal(row16.float().cpu(), sd["image_encoder.blocks.0.attn.qkv.bias"].half().float())   # padded tokens: original bias
    assert not torch.equal(pred._wv["image_encoder.blocks.0.attn.qkv.bias"].cpu(), sd["image_encoder.blocks.0.attn.qkv.bias"])
    monkeypatch.setenv("SAMPT_VIT_BIAS_CORR", "0")
    plain = _sam("vit_b", "f16", max_batch=1)
    assert not plain.bias_correction
    e_plain = as_img(plain.encode_frames(frames.to(dev)))
    monkeypatch.delenv("SAMPT_VIT_BIAS_CORR")
    print(f"\n[bias correction] ViT-B embedding rms error vs oracle: plain fp16 {rms(e_plain):.3e}, corrected {rms(e_corr):.3e}")
    assert rms(e_corr) < 0.8 * rms(e_plain)
    # history independence: another predictor, another geometry first (its own calibration), then this one
    other = _sam("vit_b", "f16", max_batch=1)
    sq, _ = synthetic_clip(T=1, H=1024, W=1024, seed=4, disc_r=60)
    other.encode_frames(sq.to(dev))
    e2 = as_img(other.encode_frames(frames.to(dev)))
    assert other.stats["bias_calibrations"] == 2 and torch.equal(e2, e_corr)
    e3 = as_img(other.encode_frames(frames.to(dev)))                      # and back and forth without recalibrating
    other.encode_frames(sq.to(dev))
    assert other.stats["bias_calibrations"] == 2 and torch.equal(e3, e_corr)


def _bias_case_frames(kind):
    g = torch.Generator().manual_seed(31)
    if kind == "black":
        return torch.zeros(1, 3, 576, 1024, dtype=torch.uint8)
    if kind == "near_black":
        return torch.randint(0, 6, (1, 3, 576, 1024), generator=g, dtype=torch.uint8)
    if kind == "flat_grey":
        return torch.full((1, 3, 576, 1024), 128, dtype=torch.uint8)
    if kind == "low_contrast":
        f, _ = synthetic_clip(T=1, H=576, W=1024, seed=9, disc_r=60)
        return (118 + f.float() * (20.0 / 255.0)).round().to(torch.uint8)
    if kind == "square_1024":
        return synthetic_clip(T=1, H=1024, W=1024, seed=4, disc_r=90)[0]
    if kind == "portrait":
        return synthetic_clip(T=1, H=1024, W=576, seed=6, disc_r=60)[0]
    return synthetic_clip(T=1, H=576, W=1024, seed=9, disc_r=60)[0]           # "outlier_weights": the bench-like frame


@pytest.mark.parametrize("kind", ["black", "near_black", "flat_grey", "low_contrast", "square_1024", "portrait", "outlier_weights"])
def test_vit_b_f16_bias_correction_off_distribution(dev, monkeypatch, kind):
    """VERDICT r5 weak #1c / ADVICE r5: the static bias correction is calibrated on ONE seeded uniform-noise frame per geometry.
    Frames far from that — black, near-black, flat, low-contrast — other geometries and weights with massive-activation channels
    must not come out WORSE than without the correction (embedding rms error vs the fp32 oracle, corrected <= 1.02 x plain)."""
    from oracle import sam_ref as R
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_b"]
    sd = {k: v.clone() for k, v in init_sam_state_dict(cfg, 72).items()}
    if kind == "outlier_weights":
        e = "image_encoder."
        sd[e + "pos_embed"][..., [5, 130, 131, 700]] += torch.tensor([1500.0, -1200.0, 900.0, -1500.0])
        sd[e + "blocks.3.norm1.weight"][40] = 30.0
        sd[e + "blocks.7.norm2.weight"][300] = -25.0
        sd[e + "blocks.5.mlp.lin2.bias"][77] = 400.0
    frames = _bias_case_frames(kind)
    ref = R.image_encoder(sd, cfg, R.preprocess(cfg, frames.float()))
    rms = lambda e: float(((e - ref).double().pow(2).mean().sqrt()) / ref.double().pow(2).mean().sqrt())
    as_img = lambda f: f.view(1, 64, 64, 256).permute(0, 3, 1, 2).cpu()
    errs = {}
    for corr in ("1", "0"):
        monkeypatch.setenv("SAMPT_VIT_BIAS_CORR", corr)
        pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f16", max_batch=1).to(dev))
        assert pred.bias_correction == (corr == "1")
        errs[corr] = rms(as_img(pred.encode_frames(frames.to(dev))))
    print(f"\n[bias correction, {kind}] embedding rms error vs oracle: plain {errs['0']:.3e}, corrected {errs['1']:.3e}, "
          f"ratio {errs['1'] / errs['0']:.3f}")
    assert errs["1"] <= 1.02 * errs["0"]


@pytest.mark.parametrize("variant,hw", [("vit_test", (100, 256)), ("vit_b", (576, 1024))])
def test_vit_f16_rel_pos_operand_images_are_exact(dev, monkeypatch, variant, hw):
    """The fp16 attention kernel takes the decomposed rel-pos tables as host-packed MFMA operand images (pack.rel_pos_operand_images,
    FlashPad::rel_ops) instead of converting the f32 tables in every workgroup: the same halves, so the embeddings are the same bits."""
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS
    cfg = SAM_CONFIGS[variant]
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (2, 3) + hw, generator=g, dtype=torch.uint8).to(dev)
    embs = {}
    for ops in ("1", "0"):
        monkeypatch.setenv("SAMPT_ATTN_REL_OPS", ops)
        pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f16", max_batch=2).to(dev))
        embs[ops] = pred.encode_frames(frames).clone()
    assert torch.equal(embs["1"], embs["0"])


@pytest.mark.parametrize("variant,precision,hw,T", [("vit_test", "f32", (144, 256), 3), ("vit_test", "f16", (100, 256), 3),
                                                     ("vit_b", "f16", (576, 1024), 3), ("vit_b", "f32", (480, 1024), 1),
                                                     ("vit_test", "f16x3", (100, 256), 3), ("vit_b", "f16x3", (576, 1024), 3)])
def test_vit_dead_row_skipping_is_exact(dev, variant, precision, hw, T):
    """Landscape frames: the blocks before the first global one run on the token rows a pixel can reach, the rest comes
    from the per-geometry cache (sampt_vit_encode_live) — bit-identical to the full computation, odd batch tail included."""
    frames, _ = synthetic_clip(T=T, H=hw[0], W=hw[1], seed=11, disc_r=hw[0] // 8)
    pred = _sam(variant, precision, max_batch=2)
    assert pred.skip_dead_rows
    a = pred.encode_frames(frames.to(dev)).clone()
    assert pred._dead_cache[hw] is not None, "geometry should have frame-independent rows"
    b = pred.encode_frames(frames.flip(0).to(dev)).flip(0)     # cache reused by a second clip, other frame first
    pred.skip_dead_rows = False
    full = pred.encode_frames(frames.to(dev))
    assert torch.equal(a, full) and torch.equal(b, full)
    # portrait / square frames have nothing to skip and take the plain entry point
    pred.skip_dead_rows = True
    sq, _ = synthetic_clip(T=1, H=hw[1], W=hw[1], seed=3, disc_r=20)
    pred.encode_frames(sq.to(dev))
    assert pred._dead_cache[(hw[1], hw[1])] is None


@pytest.mark.parametrize("precision,tol", [("f32", 3e-5), ("f16x3", 3e-5), ("f16", 1e-2)])
def test_vit_b_outlier_channels_vs_oracle(dev, precision, tol):
    """Trained SAM checkpoints carry a few "massive activation" channels in the residual stream (O(10^2 - 10^3), where random
    init stays O(1)).  Seeded weights with such channels injected — a +-1500 offset on four channels of the positional
    embedding (it rides the residual stream through every block and into the neck's split-fp16 convolution), two large
    LayerNorm gains and a large MLP output bias — go through the oracle and the HIP encoder alike: the split-fp16 paths must
    stay finite and fp32-grade, the fp16 mode must degrade gracefully."""
    from oracle import sam_ref as R
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_b"]
    sd = {k: v.clone() for k, v in init_sam_state_dict(cfg, 72).items()}
    e = "image_encoder."
    sd[e + "pos_embed"][..., [5, 130, 131, 700]] += torch.tensor([1500.0, -1200.0, 900.0, -1500.0])
    sd[e + "blocks.3.norm1.weight"][40] = 30.0
    sd[e + "blocks.7.norm2.weight"][300] = -25.0
    sd[e + "blocks.5.mlp.lin2.bias"][77] = 400.0
    frames, _ = synthetic_clip(T=1, H=576, W=1024, seed=9, disc_r=60)
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision=precision, max_batch=1).to(dev))
    feats = pred.encode_frames(frames.to(dev))
    assert bool(torch.isfinite(feats).all())
    ref = R.image_encoder(sd, cfg, R.preprocess(cfg, frames.float()))
    got = feats.view(1, 64, 64, 256).permute(0, 3, 1, 2)
    err = rel_err(got, ref)
    print(f"outlier channels, {precision}: embedding rel err {err:.3g}")
    assert err < tol


def test_hq_features_with_dead_row_skipping(dev):
    """The HQ-SAM tap (first global block's output) sits after the re-expansion: identical with and without skipping."""
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    frames, _ = synthetic_clip(T=2, H=144, W=256, seed=5)
    pred = SamPredictor(SamHip("vit_test", precision="f32", seed=72, max_batch=2, hq=True).cuda())
    a = pred.encode_frames(frames.to(dev))
    pred.skip_dead_rows = False
    b = pred.encode_frames(frames.to(dev))
    assert torch.equal(a.emb, b.emb) and torch.equal(a.hq, b.hq)


def test_predict_torch_vs_oracle(dev):
    """SamPredictor.set_image / predict_torch (points; points+mask; points+box+mask) on ViT-B in exact-f32 mode."""
    from oracle import sam_ref as R
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_b"]
    sd = init_sam_state_dict(cfg, 72)
    frames, centres = synthetic_clip(T=1, H=576, W=1024, seed=9, disc_r=60)
    img = frames[0].permute(1, 2, 0).numpy()
    ref = R.SamPredictorRef(sd, cfg)
    ref.set_image(img)
    pred = _sam("vit_b", "f32", max_batch=1)
    pred.set_image(img)
    assert pred.features.shape == (1, 256, 64, 64)
    assert rel_err(pred.features, ref.features) < 1e-4
    # decode against the ORACLE's features to isolate the decoder
    pred.set_features(ref.features[0].permute(1, 2, 0).reshape(4096, 256).to(dev), (576, 1024))
    q = disc_queries(centres, n_pos=6, r=30.0)[:, 1:]
    pts = torch.as_tensor(pred.transform.apply_coords(q.numpy(), (576, 1024)), dtype=torch.float)[None]
    lab = torch.tensor([[1, 1, 1, 1, 0, 0]], dtype=torch.int)
    m0, i0, l0 = ref.predict_torch(pts, lab, None, None, False, True)
    m1, i1, l1 = pred.predict_torch(pts.to(dev), lab.to(dev), None, None, False, True)
    assert max_abs(l1, l0) < 2e-4 and max_abs(i1, i0) < 1e-4 and max_abs(m1, m0) < 2e-4
    assert iou(m1 > 0, m0 > 0) >= 1 - 1e-3
    box = torch.tensor([[[200.0, 100.0, 700.0, 500.0]]])
    m2, i2, l2 = ref.predict_torch(pts, lab, box, l0, False, True)
    m3, i3, l3 = pred.predict_torch(pts.to(dev), lab.to(dev), box.to(dev), l0.to(dev), False, True)
    assert max_abs(l3, l2) < 2e-4 and max_abs(i3, i2) < 1e-4
    assert iou(m3 > 0, m2 > 0) >= 1 - 1e-3
    m4, i4, l4 = ref.predict_torch(pts[:, :4], lab[:, :4], None, l0, False, True)
    m5, i5, l5 = pred.predict_torch(pts[:, :4].to(dev), lab[:, :4].to(dev), None, l0.to(dev), False, True)
    assert max_abs(l5, l4) < 2e-4 and iou(m5 > 0, m4 > 0) >= 1 - 1e-3


@pytest.mark.parametrize("neg", [0, 2])
def test_sampt_end_to_end_vs_oracle(dev, neg):
    """Whole SamPt.forward on the reduced SAM geometry + full PIPS: our fused device path vs our SamPt host logic
    driving the CPU oracle predictor/tracker call by call (the reference protocol)."""
    from oracle import pips_ref as PO
    from oracle import sam_ref as R
    from sam_pt_amd.point_tracker import PipsPointTracker, PointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS, init_pips_state_dict, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd, psd = init_sam_state_dict(cfg, 72), init_pips_state_dict(72)
    frames, centres = synthetic_clip(T=10, H=128, W=256, seed=72)
    npos = 4
    q = disc_queries(centres, n_pos=npos + neg, r=9.0)
    if neg:
        q[npos:, 1:] += torch.tensor([40.0, 30.0])
    q2 = disc_queries(centres, n_pos=npos + neg, r=5.0)
    q2[:, 1:] += torch.tensor([-60.0, 20.0])
    qp = torch.stack([q, q2])                                          # 2 objects
    video = {"image": [f for f in frames], "target_hw": (128, 256), "query_points": qp}
    kw = dict(sam_iou_threshold=-1e9, positive_points_per_mask=npos, negative_points_per_mask=neg,
              iterative_refinement_iterations=3, point_tracker_mask_batch_size=5)

    class OracleTracker(PointTracker):
        def forward(self, rgbs, query_points):
            return PO.PipsTrackerRef(psd).forward(rgbs.cpu(), query_points.cpu())

    ref_model = SamPt(OracleTracker(), R.SamPredictorRef(sd, cfg), **kw).eval()
    ref = ref_model(video)
    ours = SamPt(PipsPointTracker(state_dict=psd), SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").cuda()),
                 **kw).eval()
    out = ours({**video, "image": [f.to(dev) for f in frames]})
    assert (out["visibilities"] == ref["visibilities"]).all()
    assert (out["trajectories"].round() == ref["trajectories"].round()).all()
    for m in range(2):
        a, b = out["logits"][m].cpu(), ref["logits"][m]
        fin = torch.isfinite(b)
        assert (torch.isfinite(a) == fin).all()
        for t in range(10):
            assert iou(a[t] > 0, b[t] > 0) >= 1 - 1e-3, f"mask IoU object {m} frame {t}"
    assert np.allclose(np.array(out["scores_per_frame"]), np.array(ref["scores_per_frame"]), atol=1e-3)


# ------------------------------------------------------------------------------------------ golden fixtures
def test_hip_tracker_vs_reference_golden(dev, pips_sd, clip):
    """HIP PipsPointTracker against the committed output of the REFERENCE's PipsPointTracker (tests/golden)."""
    import os
    from oracle.make_golden import golden_queries
    from sam_pt_amd.point_tracker import PipsPointTracker
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pips_tracker.npz"))
    frames, centres = clip
    q = golden_queries(centres)
    tr, vi = PipsPointTracker(state_dict=pips_sd)(frames[None].to(dev), q.to(dev))
    assert np.array_equal(vi.cpu().numpy(), g["vis"])
    assert np.abs(tr.cpu().numpy() - g["traj"]).max() < 5e-3
    assert np.array_equal(np.round(tr.cpu().numpy()), np.round(g["traj"]))


def test_hip_sam_vs_hf_golden(dev):
    """HIP decoder (exact fp32) against the committed HuggingFace SamModel outputs for the same weights."""
    import os
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sam_hf.npz"))
    cfg = SAM_CONFIGS["vit_test"]
    pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f32").cuda())
    emb = torch.from_numpy(g["emb"])                                   # (1,256,16,16) from HF
    pred.set_features(emb[0].permute(1, 2, 0).reshape(256, 256).contiguous().to(dev), (256, 256))
    pts, lab = torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["lab"]).int().to(dev)
    m, iou, low = pred.predict_torch(pts, lab, None, None, False, True)
    assert np.abs(low.cpu().numpy()[0] - g["low"]).max() < 2e-4 and np.abs(iou.cpu().numpy() - g["iou"]).max() < 1e-4
    box = torch.from_numpy(g["box"]).to(dev)
    m2, iou2, low2 = pred.predict_torch(pts, lab, box[None], torch.from_numpy(g["low"])[None].to(dev), False, True)
    assert np.abs(low2.cpu().numpy()[0] - g["low2"]).max() < 2e-4 and np.abs(iou2.cpu().numpy() - g["iou2"]).max() < 1e-4


# ------------------------------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("neg", [0, 1])
def test_ragged_prompts_fused_vs_stepwise(dev, neg):
    """Ragged prompts: frames with different numbers of visible points (incl. none -> -inf mask, a single point, points
    marked OUTSIDE_FRAME) and 3 objects feeding each other's positives as negatives.  The batched device-side chain
    (groups by prompt shape) must equal the reference call-by-call protocol run with the SAME HIP predictor."""
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS
    cfg = SAM_CONFIGS["vit_test"]
    T, M, P = 6, 3, 3 + neg
    frames, centres = synthetic_clip(T=T, H=128, W=256, seed=3)
    g = torch.Generator().manual_seed(7)
    traj = torch.rand(T, M, P, 2, generator=g) * torch.tensor([250.0, 120.0]) + 3.0
    vis = torch.ones(T, M, P)
    vis[1, :, :] = 0                       # nothing visible in frame 1 -> empty prompts: -inf logits, -inf scores
    vis[2, 1, 1:] = 0                      # a single visible point
    vis[3, 2, 0] = -2                      # OUTSIDE_FRAME code is not a visible point (sam_pt.py:734-735)
    vis[4, :, :] = torch.tensor([[1, 0, 1], [0, 1, 1], [1, 1, 0]])[:, :P] if P == 3 else vis[4]
    pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f32", max_decode_batch=4).to(dev))   # forces batch splits
    model = SamPt(PipsPointTracker(seed=72), pred, sam_iou_threshold=0.0, positive_points_per_mask=3,
                  negative_points_per_mask=neg, iterative_refinement_iterations=2).eval()
    images = frames.to(dev)
    feats = pred.encode_frames(images)
    s_f, l_f, spf_f = model._apply_sam_to_trajectories(images, traj, vis, feats)          # fused, batched
    s_s, l_s, spf_s = model._apply_sam_to_trajectories(images, traj, vis, None)           # set_image / predict_torch
    assert torch.equal(torch.isfinite(l_f).cpu(), torch.isfinite(l_s))
    assert not torch.isfinite(l_f[:, 1]).any() and (spf_f[1] == -float("inf")).all()
    fin = torch.isfinite(l_s)
    assert fin.any() and max_abs(l_f.cpu()[fin], l_s[fin]) < 2e-3
    for m in range(M):
        for t in range(T):
            assert iou(l_f[m, t].cpu() > 0, l_s[m, t] > 0) >= 1 - 1e-3
    assert np.allclose(spf_f.numpy(), spf_s.numpy(), atol=1e-4)


def test_tracker_short_clip_and_late_queries(dev, pips_sd):
    """Clips shorter than one PIPS window (tail frames repeated, pips/tracker.py:73-78), a query on the last frame
    (never anchors a window) and a single point."""
    from oracle import pips_ref as O
    from sam_pt_amd.point_tracker import PipsPointTracker
    frames, centres = synthetic_clip(T=5, H=128, W=256, seed=21)
    q = torch.cat([disc_queries(centres, n_pos=2, r=8.0, t=0), disc_queries(centres, n_pos=1, r=4.0, t=4),
                   disc_queries(centres, n_pos=1, r=2.0, t=2)])[None]
    tr_ref, vi_ref = O.PipsTrackerRef(pips_sd).forward(frames[None], q)
    tr, vi = PipsPointTracker(state_dict=pips_sd)(frames[None].to(dev), q.to(dev))
    assert (vi.cpu() == vi_ref).all() and (tr.cpu().round() == tr_ref.round()).all() and max_abs(tr, tr_ref) < 5e-3
    q1 = disc_queries(centres, n_pos=1, r=0.0, t=1)[None]
    tr_ref, vi_ref = O.PipsTrackerRef(pips_sd).forward(frames[None], q1)
    tr, vi = PipsPointTracker(state_dict=pips_sd)(frames[None].to(dev), q1.to(dev))
    assert (vi.cpu() == vi_ref).all() and (tr.cpu().round() == tr_ref.round()).all()


@pytest.mark.parametrize("name", ["reinit_median", "qmasks"])
def test_reinit_and_query_masks_device_path_vs_reference_golden(dev, pips_sd, clip, name):
    """Rows f2/f3: point re-initialisation and query_masks mode through the fused device path (cached embeddings,
    batched decoder) against the REFERENCE SamPt's committed outputs."""
    import os
    from oracle.make_golden import query_mask_video, reinit_kwargs, reinit_video, sampt_kwargs
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampt_reinit.npz"))
    frames, centres = clip
    cfg = SAM_CONFIGS["vit_test"]
    if name == "reinit_median":
        kw, video = reinit_kwargs("reinit-at-median-of-area-diff", 0), reinit_video(frames, centres, 0)
    else:
        kw = dict(sampt_kwargs(4, 1), positive_point_selection_method="random", negative_point_selection_method="random",
                  sam_iou_threshold=-1e9)
        video = query_mask_video(frames[:8], centres)
    model = SamPt(PipsPointTracker(state_dict=pips_sd), SamPredictor(SamHip(config=cfg, seed=72, precision="f32").to(dev)),
                  **kw).eval()
    torch.manual_seed(5)
    out = model({**video, "image": [f.to(dev) for f in video["image"]]})
    assert np.array_equal(out["visibilities"].numpy(), g[f"{name}_vis"])
    assert np.array_equal(np.round(out["trajectories"].numpy()), np.round(g[f"{name}_traj"]))
    masks = torch.stack([l > 0 for l in out["logits"]]).cpu().numpy()
    ref = np.unpackbits(g[f"{name}_masks"], axis=-1)[..., :masks.shape[-1]].astype(bool)
    inter, union = (masks & ref).sum(axis=(-1, -2)), (masks | ref).sum(axis=(-1, -2))
    assert np.where(union > 0, inter / np.maximum(union, 1), 1.0).min() >= 1 - 1e-3
    assert np.allclose(np.array(out["scores_per_frame"], dtype=np.float32), g[f"{name}_scores_per_frame"], atol=1e-3,
                       equal_nan=True)


# ------------------------------------------------------------------------------------------ HQ-SAM (MaskDecoderHQ)
def test_hq_sam_vit_test_vs_oracle(dev):
    """HQ-SAM on the reduced geometry, exact fp32: ViT tap + HQ features, predict_torch (points; points+box+mask) against
    the oracle (itself pinned on HuggingFace SamHQModel, tests/test_oracle_pins.py)."""
    from oracle import sam_ref as R
    from sam_pt_amd.sam_predictor import ClipFeatures, SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72, hq=True)
    frames, centres = synthetic_clip(T=3, H=144, W=256, seed=5)
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32", max_batch=2).to(dev))   # 2 encode batches
    assert pred.model.hq
    feats = pred.encode_frames(frames.to(dev))
    assert isinstance(feats, ClipFeatures) and feats.hq.shape == (3, 64 * 64, 32)
    x = R.preprocess(cfg, frames.float())
    emb, interm = R.image_encoder(sd, cfg, x, return_interm=True)
    hq_ref = torch.cat([R.hq_features(sd, emb[i:i + 1], interm[i:i + 1]) for i in range(3)])          # (3,32,64,64)
    assert rel_err(feats.emb.view(3, 16, 16, 256).permute(0, 3, 1, 2), emb) < 3e-5
    assert rel_err(feats.hq.view(3, 64, 64, 32).permute(0, 3, 1, 2), hq_ref) < 3e-5
    ref = R.SamPredictorRef(sd, cfg, hq=True)
    img = frames[1].permute(1, 2, 0).numpy()
    ref.set_image(img)
    pred.set_features(feats[1], (144, 256))
    q = disc_queries(centres, n_pos=4, r=9.0)[:, 1:]
    pts = torch.as_tensor(pred.transform.apply_coords(q.numpy(), (144, 256)), dtype=torch.float)[None]
    lab = torch.tensor([[1, 1, 1, 0]], dtype=torch.int)
    m0, i0, l0 = ref.predict_torch(pts, lab, None, None, False, True)
    m1, i1, l1 = pred.predict_torch(pts.to(dev), lab.to(dev), None, None, False, True)
    assert max_abs(l1, l0) < 3e-4 and max_abs(i1, i0) < 1e-4 and max_abs(m1, m0) < 3e-4
    box = torch.tensor([[[20.0, 10.0, 200.0, 120.0]]])
    m2, i2, l2 = ref.predict_torch(pts, lab, box, l0, False, True)
    m3, i3, l3 = pred.predict_torch(pts.to(dev), lab.to(dev), box.to(dev), l0.to(dev), False, True)
    assert max_abs(l3, l2) < 3e-4 and max_abs(i3, i2) < 1e-4
    assert iou(m3 > 0, m2 > 0) >= 1 - 1e-3
    # and the HQ term is really there: the plain-SAM decoder on the same weights gives a different mask
    plain = R.SamPredictorRef(sd, cfg)
    plain.set_image(img)
    _, _, lp = plain.predict_torch(pts, lab, None, None, False, True)
    assert max_abs(lp, l0) > 1e-2
    # set_image path (encode + hq features for one HWC frame)
    pred.set_image(img)
    _, _, l4 = pred.predict_torch(pts.to(dev), lab.to(dev), None, None, False, True)
    assert max_abs(l4, l0) < 3e-4


def test_hq_sam_decoder_full_geometry_vs_oracle(dev):
    """HQ decoder at the production geometry (grid 64, 1024 px, vit_dim 768) from random embeddings, exact fp32."""
    from oracle import sam_ref as R
    from sam_pt_amd.sam_predictor import ClipFeatures, SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    import ctypes as C
    from sam_pt_amd import _lib
    cfg = SAM_CONFIGS["vit_b"]
    sd = init_sam_state_dict(cfg, 72, hq=True)
    g = torch.Generator().manual_seed(11)
    emb = torch.randn(1, 256, 64, 64, generator=g) * 0.5
    interm = torch.randn(1, 64, 64, 768, generator=g)
    hq_ref = R.hq_features(sd, emb, interm)
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32", max_batch=1, max_decode_batch=2).to(dev))
    pred._ensure()
    emb_d = emb[0].permute(1, 2, 0).reshape(1, 4096, 256).contiguous().to(dev)
    int_d = interm.reshape(1, 4096, 768).contiguous().to(dev)
    hq_d = torch.empty((1, 65536, 32), device=dev)
    n = C.c_size_t()
    _lib.check(pred._lib.sampt_dec_hq_workspace_bytes(pred._dec, 1, C.byref(n)), "ws")
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    _lib.check(pred._lib.sampt_dec_hq_features(pred._dec, 1, _lib.ptr(emb_d), _lib.ptr(int_d), _lib.ptr(hq_d), _lib.ptr(ws),
                                               ws.numel(), _lib.stream_ptr()), "hq")
    assert rel_err(hq_d.view(1, 256, 256, 32).permute(0, 3, 1, 2), hq_ref) < 3e-5
    ref = R.SamPredictorRef(sd, cfg, hq=True)
    ref.features, ref.hq_feat = emb, hq_ref
    ref.original_size = ref.input_size = (576, 1024)
    ref.is_image_set = True
    pred.set_features(ClipFeatures(emb_d, hq_d)[0], (576, 1024))
    pts = torch.tensor([[[300.0, 200.0], [420.0, 260.0], [800.0, 100.0]]])
    lab = torch.tensor([[1, 1, 0]], dtype=torch.int)
    m0, i0, l0 = ref.predict_torch(pts, lab, None, None, False, True)
    m1, i1, l1 = pred.predict_torch(pts.to(dev), lab.to(dev), None, None, False, True)
    assert max_abs(l1, l0) < 5e-4 and max_abs(i1, i0) < 1e-4
    assert iou(m1 > 0, m0 > 0) >= 1 - 1e-3


@pytest.mark.parametrize("neg", [0, 1])
def test_hq_sampt_fused_vs_oracle_stepwise(dev, pips_sd, neg):
    """End to end with the HQ-SAM decoder: fused device path vs the oracle predictor driven call by call."""
    from oracle import sam_ref as R
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72, hq=True)
    frames, centres = synthetic_clip(T=6, H=128, W=256, seed=3)
    q = disc_queries(centres, n_pos=3 + neg, r=9.0)
    if neg:
        q[3:, 1:] += torch.tensor([40.0, 30.0])
    video = {"image": [f for f in frames], "target_hw": (128, 256), "query_points": q[None]}
    kw = dict(sam_iou_threshold=-1e9, positive_points_per_mask=3, negative_points_per_mask=neg,
              iterative_refinement_iterations=2)
    trk = PipsPointTracker(state_dict=pips_sd)
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").to(dev))
    out = SamPt(trk, pred, **kw).eval()(video)
    ref_pred = R.SamPredictorRef(sd, cfg, hq=True)
    ref_pred.model = torch.nn.Module()
    ref_pred.model.device, ref_pred.model.mask_threshold = torch.device("cpu"), 0.0

    ref = SamPt(trk, ref_pred, **kw).eval()           # same trajectories on both sides: isolates the SAM stage
    _, l_ref, spf = ref._apply_sam_to_trajectories(frames, out["trajectories"].cpu(), out["visibilities"].cpu(), None)
    l_got = torch.stack([l for l in out["logits"]]).cpu()
    for t in range(6):
        assert iou(l_got[0, t] > 0, l_ref[0, t] > 0) >= 1 - 1e-3
    assert max_abs(l_got, l_ref) < 3e-3
    assert np.allclose(np.array(out["scores_per_frame"], dtype=np.float32).reshape(-1), spf.numpy().reshape(-1), atol=1e-4)


def test_many_objects_long_prompts(dev):
    """BASELINE config #5 prompt shape: 5 objects x 16 positive points; every other object's positives are appended as
    negatives (sam_pt.py:737-756) -> k = 80 prompt points, 88 decoder tokens in the HQ decoder.  Fused batched chain vs the
    call-by-call protocol on the same HIP predictor, plus one frame against the CPU oracle."""
    from oracle import sam_ref as R
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72, hq=True)
    T, M, P = 3, 5, 16
    frames, _ = synthetic_clip(T=T, H=128, W=256, seed=3)
    g = torch.Generator().manual_seed(17)
    traj = torch.rand(T, M, P, 2, generator=g) * torch.tensor([250.0, 120.0]) + 3.0
    vis = torch.ones(T, M, P)
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32", max_decode_batch=8).to(dev))
    model = SamPt(PipsPointTracker(seed=72), pred, sam_iou_threshold=-1e9, positive_points_per_mask=P,
                  negative_points_per_mask=0, iterative_refinement_iterations=1).eval()
    images = frames.to(dev)
    feats = pred.encode_frames(images)
    _, l_f, spf_f = model._apply_sam_to_trajectories(images, traj, vis, feats)
    _, l_s, spf_s = model._apply_sam_to_trajectories(images, traj, vis, None)
    assert max_abs(l_f.cpu(), l_s) < 2e-3 and np.allclose(spf_f.numpy(), spf_s.numpy(), atol=1e-4)
    ref_pred = R.SamPredictorRef(sd, cfg, hq=True)
    ref_pred.model = torch.nn.Module()
    ref_pred.model.device, ref_pred.model.mask_threshold = torch.device("cpu"), 0.0
    ref = SamPt(PipsPointTracker(seed=72), ref_pred, sam_iou_threshold=-1e9, positive_points_per_mask=P,
                negative_points_per_mask=0, iterative_refinement_iterations=1).eval()
    _, l_r, _ = ref._apply_sam_to_trajectories(frames[:1], traj[:1], vis[:1], None)
    assert max_abs(l_f[:, :1].cpu(), l_r) < 3e-3
    for m in range(M):
        assert iou(l_f[m, 0].cpu() > 0, l_r[m, 0] > 0) >= 1 - 1e-3


# ------------------------------------------------------------------------------------------ PIPS++ (row f4)
def test_pips2_chunk_vs_oracle(dev, clip):
    """Stride-8 encoder + one PIPS++ chunk (16 iterations of: 3-template local correlation, sincos flow embedding, 1-D
    ResNet over time) through the C ABI against the CPU oracle (bit-identical to the reference model, test_oracle_pins)."""
    from oracle import pips2_ref as O2
    from oracle import pips_ref as O1
    from sam_pt_amd.point_tracker import PipsPlusPlusPointTracker
    from sam_pt_amd.weights import init_pips2_state_dict
    frames, centres = clip
    sd = init_pips2_state_dict(72)
    trk = PipsPlusPlusPointTracker(state_dict=sd, iters=16)
    pyr = trk.compute_pyramid(frames.to(dev))
    fm = O2.fnet(sd, O1.normalize_rgbs(frames), 8)                                # (12,128,16,32)
    assert rel_err(pyr[0].permute(0, 3, 1, 2), fm) < 2e-5
    q = disc_queries(centres, n_pos=5, r=9.0)[:, 1:]
    preds, _ = O2.pips2_forward(sd, q[None].repeat(12, 1, 1), fm, iters=16)
    got = trk._track(pyr, list(range(12)), q)
    assert max_abs(got, preds[-1]) < 5e-3
    assert torch.equal(got.cpu().round(), preds[-1].round())


def test_pips2_tracker_vs_reference_golden(dev, clip):
    """HIP PipsPlusPlusPointTracker against the committed outputs of the REFERENCE's tracker (single query frame, with and
    without chunking) and, for what the reference cannot run (several query frames, a query on the last frame), against
    the oracle."""
    import os
    from oracle import pips2_ref as O2
    from sam_pt_amd.point_tracker import PipsPlusPlusPointTracker
    from sam_pt_amd.weights import init_pips2_state_dict
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pips2.npz"))
    frames, centres = clip
    sd = init_pips2_state_dict(72)
    for name, maxlen, iters in (("t0", 128, 16), ("t5", 128, 8), ("t5_chunked", 5, 3)):
        trk = PipsPlusPlusPointTracker(state_dict=sd, max_sequence_length=maxlen, iters=iters)
        q = torch.from_numpy(g[f"trk_{name}_q"])
        tr, vi = trk(frames[None].to(dev), q.to(dev))
        assert np.abs(tr.cpu().numpy() - g[f"trk_{name}_traj"]).max() < 5e-3, name
        assert np.array_equal(np.round(tr.cpu().numpy()), np.round(g[f"trk_{name}_traj"])), name
        assert np.array_equal(vi.cpu().numpy(), g[f"trk_{name}_vis"])
    trk = PipsPlusPlusPointTracker(state_dict=sd, iters=4, image_size=(128, 192))     # float video after the pre-resize
    tr, _ = trk(frames[None].to(dev), torch.from_numpy(g["trk_resized_q"]).to(dev))
    assert np.abs(tr.cpu().numpy() - g["trk_resized_traj"]).max() < 5e-3
    q = torch.cat([disc_queries(centres, n_pos=3, r=9.0, t=0), disc_queries(centres, n_pos=2, r=6.0, t=5),
                   disc_queries(centres, n_pos=1, r=3.0, t=11)])[None]
    trk = PipsPlusPlusPointTracker(state_dict=sd, iters=8)
    tr, vi = trk(frames[None].to(dev), q.to(dev))
    tr_ref, vi_ref = O2.Pips2TrackerRef(sd, 8, 128, 8).forward(frames[None], q)
    assert max_abs(tr, tr_ref) < 5e-3 and torch.equal(vi.cpu(), vi_ref)
    for i in range(q.shape[1]):                                                   # query frame holds the query point
        t = int(q[0, i, 0])
        assert torch.allclose(tr[0, t, i].cpu(), q[0, i, 1:], atol=1e-4)


def test_sampt_with_pips2_tracker(dev, clip):
    """SamPt on the device path with the PIPS++ tracker (second-stream overlap, prepare()): trajectories are the
    tracker's, masks equal the call-by-call protocol on the same trajectories."""
    from sam_pt_amd.point_tracker import PipsPlusPlusPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS
    frames, centres = clip
    cfg = SAM_CONFIGS["vit_test"]
    q = disc_queries(centres, n_pos=4, r=9.0)
    video = {"image": [f for f in frames[:8]], "target_hw": (128, 256), "query_points": q[None]}
    trk = PipsPlusPlusPointTracker(seed=72, iters=4)
    pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f32").to(dev))
    model = SamPt(trk, pred, sam_iou_threshold=-1e9, positive_points_per_mask=4, negative_points_per_mask=0,
                  iterative_refinement_iterations=2).eval()
    out = model(video)
    tr, _ = trk(frames[None, :8].to(dev), q[None].to(dev))
    assert max_abs(out["trajectories"][:, 0], tr[0]) < 1e-4
    images = frames[:8].to(dev)
    _, l_s, _ = model._apply_sam_to_trajectories(images, out["trajectories"].cpu(), out["visibilities"].cpu(), None)
    l_f = torch.stack(out["logits"]).cpu()
    assert max_abs(l_f, l_s) < 2e-3


# ------------------------------------------------------------------------------------------ limits
def test_prompt_size_limits(dev):
    """Large prompts against the oracle: 120 points (one key chunk of the image->token attention), 121 / 300 / 1700
    points (several chunks: the reference's 16 points x M objects fed to each other as negatives, sam_pt.py:737-756, and
    the VIS adapter's mask batches exceed 120); beyond SAMPT_DEC_MAX_POINTS the call is refused loudly."""
    from oracle import sam_ref as R
    from sam_pt_amd import _lib
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    frames, _ = synthetic_clip(T=1, H=144, W=256, seed=5)
    img = frames[0].permute(1, 2, 0).numpy()
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").to(dev))
    ref = R.SamPredictorRef(sd, cfg)
    pred.set_image(img), ref.set_image(img)
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(1, 120, 2, generator=g) * torch.tensor([250.0, 140.0])).float()
    lab = (torch.rand(1, 120, generator=g) > 0.3).int()
    box = torch.tensor([[[20.0, 10.0, 200.0, 120.0]]])
    m0, i0, l0 = ref.predict_torch(pts, lab, box, None, False, True)
    m1, i1, l1 = pred.predict_torch(pts.to(dev), lab.to(dev), box.to(dev), None, False, True)
    assert max_abs(l1, l0) < 5e-4 and max_abs(i1, i0) < 1e-4
    for k in (121, 300, 1700):
        pts2 = (torch.rand(1, k, 2, generator=g) * torch.tensor([250.0, 140.0])).float()
        lab2 = (torch.rand(1, k, generator=g) > 0.3).int()
        m2, i2, l2 = ref.predict_torch(pts2, lab2, box, l0, False, True)
        m3, i3, l3 = pred.predict_torch(pts2.to(dev), lab2.to(dev), box.to(dev), l0.to(dev), False, True)
        assert max_abs(l3, l2) < 1e-3 and max_abs(i3, i2) < 2e-4, k
        assert iou(m3 > 0, m2 > 0) >= 1 - 1e-3
    pts4 = torch.zeros(1, 4100, 2)
    with pytest.raises(_lib.SamptError):
        pred.predict_torch(pts4.to(dev), torch.ones(1, 4100, dtype=torch.int).to(dev), box.to(dev), None, False, True)


def test_single_frame_and_single_point_video(dev, pips_sd):
    """Degenerate clips: T = 1 (no tracker window ever runs: the trajectory is the query) and one point."""
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS
    cfg = SAM_CONFIGS["vit_test"]
    frames, centres = synthetic_clip(T=1, H=128, W=256, seed=3)
    q = disc_queries(centres, n_pos=1, r=0.0)
    pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f32").to(dev))
    model = SamPt(PipsPointTracker(state_dict=pips_sd), pred, sam_iou_threshold=-1e9, positive_points_per_mask=1,
                  negative_points_per_mask=0, iterative_refinement_iterations=1).eval()
    out = model({"image": [frames[0]], "target_hw": (128, 256), "query_points": q[None]})
    assert out["trajectories"].shape == (1, 1, 1, 2) and torch.allclose(out["trajectories"][0, 0, 0].cpu(), q[0, 1:])
    assert out["logits"][0].shape == (1, 128, 256) and torch.isfinite(out["logits"][0]).all()


def test_predict_torch_multimask_vs_oracle(dev):
    """multimask_output=True (not on the SAM-PT path, part of the SamPredictor surface): the three masks / IoUs of mask
    tokens 1..3 against the oracle (whose multimask branch agrees with HuggingFace SamModel to 3e-5)."""
    from oracle import sam_ref as R
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    frames, centres = synthetic_clip(T=1, H=144, W=256, seed=5)
    img = frames[0].permute(1, 2, 0).numpy()
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").to(dev))
    ref = R.SamPredictorRef(sd, cfg)
    pred.set_image(img), ref.set_image(img)
    q = disc_queries(centres, n_pos=3, r=9.0)[:, 1:]
    pts = torch.as_tensor(pred.transform.apply_coords(q.numpy(), (144, 256)), dtype=torch.float)[None]
    lab = torch.tensor([[1, 1, 0]], dtype=torch.int)
    m0, i0, l0 = ref.predict_torch(pts, lab, None, None, True, True)
    m1, i1, l1 = pred.predict_torch(pts.to(dev), lab.to(dev), None, None, True, True)
    assert m1.shape == (1, 3, 144, 256) and i1.shape == (1, 3) and l1.shape == (1, 3, 64, 64)
    assert max_abs(l1, l0) < 3e-4 and max_abs(i1, i0) < 1e-4 and max_abs(m1, m0) < 3e-4
    box = torch.tensor([[[20.0, 10.0, 200.0, 120.0]]])
    m2, i2, l2 = ref.predict_torch(pts, lab, box, l0[:, :1], True, False)
    m3, i3, l3 = pred.predict_torch(pts.to(dev), lab.to(dev), box.to(dev), l0[:, :1].to(dev), True, False)
    assert m3.dtype == torch.bool and max_abs(l3, l2) < 3e-4 and max_abs(i3, i2) < 1e-4
    assert all(iou(m3[0, j].cpu(), m2[0, j]) >= 1 - 1e-3 for j in range(3))
    # the single-mask call afterwards is unaffected by the multimask one
    m4, i4, l4 = pred.predict_torch(pts.to(dev), lab.to(dev), None, None, False, True)
    m5, i5, l5 = ref.predict_torch(pts, lab, None, None, False, True)
    assert max_abs(l4, l5) < 3e-4 and m4.shape == (1, 1, 144, 256)


def test_set_image_with_resize_vs_pil_and_oracle(dev):
    """SamPredictor.set_image on a frame whose longest side is not img_size: the device resize is bit-identical to
    PIL.Image.resize(BILINEAR) (what upstream's ResizeLongestSide.apply_image does), and prediction at the original
    resolution (input_size != original_size in postprocess_masks) matches the oracle."""
    from PIL import Image
    from oracle import sam_ref as R
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").to(dev))
    rng = np.random.default_rng(1)
    for h, w in ((60, 107), (300, 200), (256, 100)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        nh, nw = pred.transform.get_preprocess_shape(h, w, cfg.img_size)
        got = pred.transform.apply_image_torch(torch.as_tensor(img, device=dev))
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert got.shape == (nh, nw, 3) and np.array_equal(got.cpu().numpy(), ref), (h, w)
    frames, centres = synthetic_clip(T=1, H=144, W=256, seed=5)
    img = np.ascontiguousarray(frames[0].permute(1, 2, 0).numpy()[::2, ::2])      # 72 x 128 -> resized to 144 x 256
    ref_p = R.SamPredictorRef(sd, cfg)
    ref_p.set_image(img), pred.set_image(img)
    assert pred.original_size == (72, 128) and pred.input_size == (144, 256) == ref_p.input_size
    assert rel_err(pred.features, ref_p.features) < 3e-5
    pts_np = np.array([[40.0, 30.0], [80.0, 50.0], [100.0, 20.0]])
    lab_np = np.array([1, 1, 0])
    pc = torch.as_tensor(ref_p.transform.apply_coords(pts_np, ref_p.original_size), dtype=torch.float)[None]
    m0, i0, l0 = ref_p.predict_torch(pc, torch.as_tensor(lab_np)[None].int(), None, None, False, True)
    m1, i1, l1 = pred.predict(pts_np, lab_np, multimask_output=False, return_logits=True)      # numpy flavour
    assert m1.shape == (1, 72, 128) and np.abs(m1 - m0[0].numpy()).max() < 3e-4 and np.abs(i1 - i0[0].numpy()).max() < 1e-4


def test_arbitrary_frame_sizes(dev, pips_sd):
    """Frames whose sides are not multiples of 32 and whose longest side is not the SAM input size (e.g. native 480 x 854):
    the tracker runs at frame resolution (floor-divided feature maps / pyramid like the reference), SAM resizes with the
    PIL-exact kernel, masks come back at frame resolution.  Tracker vs oracle; fused device path vs the call-by-call
    protocol; the latter vs the oracle predictor driven the same way."""
    from oracle import pips_ref as O
    from oracle import sam_ref as R
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    frames, centres = synthetic_clip(T=9, H=120, W=214, seed=21)
    q = disc_queries(centres, n_pos=4, r=8.0)[None]
    tr_ref, vi_ref = O.PipsTrackerRef(pips_sd).forward(frames[None], q)
    trk = PipsPointTracker(state_dict=pips_sd)
    tr, vi = trk(frames[None].to(dev), q.to(dev))
    assert (vi.cpu() == vi_ref).all() and (tr.cpu().round() == tr_ref.round()).all() and max_abs(tr, tr_ref) < 5e-3
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").to(dev))
    kw = dict(sam_iou_threshold=-1e9, positive_points_per_mask=4, negative_points_per_mask=0, iterative_refinement_iterations=2)
    model = SamPt(trk, pred, **kw).eval()
    video = {"image": [f for f in frames], "target_hw": (120, 214), "query_points": q}
    out = model(video)
    assert out["logits"][0].shape == (9, 120, 214)
    images = frames.to(dev)
    _, l_s, s_s = model._apply_sam_to_trajectories(images, out["trajectories"].cpu(), out["visibilities"].cpu(), None)
    l_f = torch.stack(out["logits"]).cpu()
    assert max_abs(l_f, l_s) < 2e-3
    ref_pred = R.SamPredictorRef(sd, cfg)
    ref_pred.model = torch.nn.Module()
    ref_pred.model.device, ref_pred.model.mask_threshold = torch.device("cpu"), 0.0
    ref = SamPt(trk, ref_pred, **kw).eval()
    _, l_r, _ = ref._apply_sam_to_trajectories(frames, out["trajectories"].cpu(), out["visibilities"].cpu(), None)
    assert max_abs(l_s, l_r) < 3e-3
    for t in range(9):
        assert iou(l_f[0, t] > 0, l_r[0, t] > 0) >= 1 - 1e-3


def test_predict_torch_prompt_batch_equals_single_calls(dev):
    """Upstream's batch dimension of predict_torch (B prompts against one image, what the automatic mask generator
    issues): row b of a batched call is bit-identical to the single call with prompt b, for both decoder flavours."""
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    frames, _ = synthetic_clip(T=1, H=96, W=128, seed=9)
    pred = SamPredictor(SamHip(config=cfg, state_dict=init_sam_state_dict(cfg, 72), precision="f32").to(dev))
    pred.set_image(frames[0].permute(1, 2, 0).numpy())
    g = torch.Generator().manual_seed(2)
    B = 5
    pts = (torch.rand(B, 2, 2, generator=g) * torch.tensor([256.0, 192.0])).to(dev)
    lab = torch.tensor([[1, 0]] * B, dtype=torch.int, device=dev)
    for multi in (True, False):
        m, i, l = pred.predict_torch(pts, lab, multimask_output=multi, return_logits=True)
        nm = 3 if multi else 1
        assert m.shape == (B, nm, 96, 128) and i.shape == (B, nm) and l.shape == (B, nm, 64, 64)
        for b in range(B):
            mb, ib, lb = pred.predict_torch(pts[b:b + 1], lab[b:b + 1], multimask_output=multi, return_logits=True)
            assert torch.equal(mb[0], m[b]) and torch.equal(ib[0], i[b]) and torch.equal(lb[0], l[b])
    assert not torch.equal(m[0], m[1])


def test_automatic_mask_generator_hip_vs_oracle(dev):
    """SamAutomaticMaskGenerator (VIS adapter, row f3) on the HIP predictor: every record is one of SAM's three masks for
    its grid point, and the record set agrees with the same generator driven by the CPU oracle predictor."""
    from oracle import sam_ref as R
    from sam_pt_amd import automatic_mask_generator as A
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    frames, _ = synthetic_clip(T=1, H=96, W=128, seed=3)
    img = frames[0].permute(1, 2, 0).contiguous().numpy()
    kw = dict(points_per_side=4, points_per_batch=6, pred_iou_thresh=0.0, stability_score_thresh=0.0)
    sam = SamHip(config=cfg, state_dict=sd, precision="f32").to(dev)
    ours = A.SamAutomaticMaskGenerator(sam, **kw).generate(img)
    ref = A.SamAutomaticMaskGenerator(None, predictor=R.SamPredictorRef(sd, cfg), **kw).generate(img)
    assert len(ours) > 0 and abs(len(ours) - len(ref)) <= max(1, len(ref) // 10)
    ious_ = [r["predicted_iou"] for r in ours]
    assert ious_ == sorted(ious_, reverse=True)
    matched = 0
    for r in ours:
        assert r["segmentation"].shape == (96, 128) and r["area"] == int(r["segmentation"].sum())
        cands = [q for q in ref if q["point_coords"] == r["point_coords"] and abs(q["predicted_iou"] - r["predicted_iou"]) < 1e-3]
        if cands:
            matched += 1
            # same mask up to the pixels whose logit sits within the fp32 noise of 0 (small masks: a pixel or two)
            diff = min(int((r["segmentation"] ^ q["segmentation"]).sum()) for q in cands)
            assert diff <= max(3, 0.01 * r["area"]), (diff, r["area"])
    assert matched >= 0.9 * len(ours)
    # one crop layer + small-region clean-up run end to end on the device path
    more = A.SamAutomaticMaskGenerator(sam, points_per_side=2, points_per_batch=8, pred_iou_thresh=0.0,
                                       stability_score_thresh=0.0, crop_n_layers=1, crop_n_points_downscale_factor=2,
                                       min_mask_region_area=6).generate(img)
    assert more and all(r["segmentation"].shape == (96, 128) for r in more)


@pytest.mark.parametrize("f16x3", ["0", "1"])
def test_decoder_image_side_gemms_both_arithmetics(dev, monkeypatch, f16x3):
    """The decoder's GEMMs over the image tokens (fused K|V|Q' projections with the folded positional term, out-projection,
    the two transposed convolutions) run as 3-term split-fp16 MFMAs by default and as exact-f32 MFMAs with
    SAMPT_DEC_F16X3=0 (row-mapped transposed convolutions, broadcast residual in gemm_f32): same tolerance against the
    oracle either way."""
    from oracle import sam_ref as R
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    monkeypatch.setenv("SAMPT_DEC_F16X3", f16x3)
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    frames, centres = synthetic_clip(T=1, H=144, W=256, seed=5)
    img = frames[0].permute(1, 2, 0).numpy()
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").to(dev))
    ref = R.SamPredictorRef(sd, cfg)
    pred.set_image(img), ref.set_image(img)
    q = disc_queries(centres, n_pos=3, r=9.0)[:, 1:]
    pts = torch.as_tensor(pred.transform.apply_coords(q.numpy(), (144, 256)), dtype=torch.float)[None]
    lab = torch.tensor([[1, 1, 0]], dtype=torch.int)
    m0, i0, l0 = ref.predict_torch(pts, lab, None, None, False, True)
    m1, i1, l1 = pred.predict_torch(pts.to(dev), lab.to(dev), None, None, False, True)
    assert max_abs(l1, l0) < 3e-4 and max_abs(i1, i0) < 1e-4 and max_abs(m1, m0) < 3e-4


# ------------------------------------------------------------------------------------------ streams / hipGraph
def test_track_decode_graph_replay_is_bitwise_eager(dev):
    """north_star "hipGraph capture of the per-frame decode": the chain of sampt_sam_track_decode replayed from a captured
    hipGraph (persistent bucket buffers, non-default stream) gives the eager chain's results bit for bit; first call of a
    signature is eager, the second captures, later ones only replay; new inputs in the same buffers are picked up."""
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS
    cfg = SAM_CONFIGS["vit_test"]
    pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f32", max_decode_batch=4).to(dev))
    frames, _ = synthetic_clip(T=4, H=128, W=256, seed=3)
    feats = pred.encode_frames(frames.to(dev))
    g = torch.Generator().manual_seed(5)
    F_, K = 4, 5
    st = pred.decode_staging(F_, K, (128, 256))
    side = torch.cuda.Stream(device=dev)

    def fill(seed):
        gg = torch.Generator().manual_seed(seed)
        st["feats"].copy_(feats[torch.randperm(4, generator=gg)])
        st["pts"].copy_((torch.rand(F_, K, 2, generator=gg) * torch.tensor([250.0, 120.0])).to(dev))
        st["labels"].copy_((torch.rand(F_, K, generator=gg) > 0.3).int().to(dev))
        st["k_item"].copy_(torch.tensor([5, 3, 1, 4], dtype=torch.int32).to(dev))

    def run(graph, out_l, out_s):
        pred.track_decode(st["feats"], st["pts"], st["labels"], K, -1, 3, 0.0, (128, 256), out_l, out_s,
                          k_item=st["k_item"], graph=graph)

    c0 = pred.graph_stats()
    for it, seed in enumerate((1, 2, 3, 4)):
        fill(seed)
        torch.cuda.synchronize()
        ref_l, ref_s = torch.empty_like(st["logits"]), torch.empty_like(st["score"])
        run(False, ref_l, ref_s)                                         # eager, default stream, fresh outputs
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            run(True, st["logits"], st["score"])
        side.synchronize()
        assert torch.equal(st["logits"], ref_l) and torch.equal(st["score"], ref_s), f"call {it}"
    c1 = pred.graph_stats()
    assert c1[1] - c0[1] == 1 and c1[2] - c0[2] == 3, (c0, c1)           # 1 capture; replays on calls 2, 3, 4


def test_pipelined_decoder_stream_equals_serial(dev, pips_sd):
    """SamPt's three-stream schedule (decoder chains per encoder batch on their own stream, replayed from hipGraphs) vs the
    serial schedule: same trajectories, masks within the batched-vs-unbatched tolerance."""
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS
    cfg = SAM_CONFIGS["vit_test"]
    frames, centres = synthetic_clip(T=11, H=128, W=256, seed=72)
    q = torch.stack([disc_queries(centres, n_pos=4, r=9.0), disc_queries(centres, n_pos=4, r=5.0) + torch.tensor([0.0, -50.0, 20.0])])
    outs = []
    for pipelined in (False, True):
        pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f32", max_batch=4, max_decode_batch=8).to(dev))
        model = SamPt(PipsPointTracker(state_dict=pips_sd), pred, sam_iou_threshold=-1e9, positive_points_per_mask=4,
                      negative_points_per_mask=0, iterative_refinement_iterations=3).eval()
        model.pipeline_decoder = model.overlap_tracker_encoder_fnet = pipelined
        for rep in range(3 if pipelined else 1):                          # repeats exercise capture and replay
            out = model({"image": [f.to(dev) for f in frames], "target_hw": (128, 256), "query_points": q})
        torch.cuda.synchronize()
        outs.append(out)
        if pipelined:
            assert pred.graph_stats()[2] > 0, "the pipelined schedule never replayed a graph"
    a, b = outs
    assert torch.equal(a["trajectories"], b["trajectories"]) and torch.equal(a["visibilities"], b["visibilities"])
    for m in range(2):
        assert max_abs(a["logits"][m], b["logits"][m]) < 2e-3
        for t in range(11):
            assert iou(a["logits"][m][t] > 0, b["logits"][m][t] > 0) >= 1 - 1e-3
    assert np.allclose(np.array(a["scores_per_frame"]), np.array(b["scores_per_frame"]), atol=1e-4)


# ------------------------------------------------------------------------------------------ unchanged-SamPt fast path
def test_reference_protocol_with_clip_embedding_prefetch_is_exact(dev, pips_sd, monkeypatch):
    """SURVEY.md §7.1 / VERDICT r2 #4: under the reference protocol (tracker first, then per frame ``set_image(numpy)`` +
    sequential ``predict_torch``; sam_pt.py:584-593, 848-858) ``set_image`` resolves to the embedding of the clip the tracker
    was given, batch-encoded once.  Same logits bit for bit as with per-frame encoding; a frame that is not in the clip, or a
    clip edited in place after the tracker saw it, takes the normal path."""
    import bench
    from sam_pt_amd import prefetch
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS
    cfg = SAM_CONFIGS["vit_test"]
    frames, centres = synthetic_clip(T=6, H=128, W=256, seed=3)
    q = disc_queries(centres, n_pos=4, r=9.0)[None]
    kw = dict(sam_iou_threshold=-1e9, positive_points_per_mask=4, negative_points_per_mask=0, iterative_refinement_iterations=2)
    video = {"image": [f.to(dev) for f in frames], "target_hw": (128, 256), "query_points": q}
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SAMPT_PREFETCH", mode)
        prefetch.clear()
        for k in prefetch.stats:
            prefetch.stats[k] = 0
        pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f16", max_batch=4).to(dev))
        model = SamPt(PipsPointTracker(state_dict=pips_sd), bench.ReferenceApiPredictor(pred), **kw).eval()
        outs[mode] = model(video)
        if mode == "1":
            # query-mask pass (sam_pt.py:181): 1 set_image before the tracker has published anything -> a miss; then 6 hits
            assert prefetch.stats["hits"] == 6 and prefetch.stats["clips_encoded"] == 1, prefetch.stats
            assert pred.stats["encoded_frames"] <= 6 + 1 + 1, pred.stats      # clip + query frame (+ dead-row cache build)
            assert prefetch._get_current() is None, "the clip stays pinned after its last frame was served"
            again = model(video)                                           # the same clip again: encoded again
            assert prefetch.stats["clips_encoded"] == 2 and prefetch.stats["hits"] == 12, prefetch.stats
            assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(again["logits"], outs[mode]["logits"]))
            other = (frames[0].permute(1, 2, 0).numpy().copy())
            other[5, 7, 1] ^= 1                                            # one bit off: not a frame of the clip
            h0 = prefetch.stats["hits"]
            pred.set_image(other)
            assert prefetch.stats["hits"] == h0
        else:
            assert prefetch.stats["hits"] == 0
    for a, b in zip(outs["0"]["logits"], outs["1"]["logits"]):
        assert torch.equal(a.cpu(), b.cpu())
    assert torch.equal(outs["0"]["trajectories"], outs["1"]["trajectories"])


# ------------------------------------------------------------------------------------------ clips in flight
def test_stream_of_clips_equals_forward(dev, pips_sd):
    """``SamPt.stream`` (forward_begin of clip i + 1 before forward_end of clip i: the decoder chain of one clip runs beside
    the tracker encoder / first encoder batches of the next) returns, clip by clip, exactly what ``forward`` returns: same
    kernels on the same streams, only the host's waiting point moves — so bitwise, not within a tolerance."""
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS
    cfg = SAM_CONFIGS["vit_test"]
    videos = []
    for seed, T in ((72, 11), (73, 9), (74, 11), (75, 5)):
        frames, centres = synthetic_clip(T=T, H=128, W=256, seed=seed)
        q = torch.stack([disc_queries(centres, n_pos=4, r=9.0), disc_queries(centres, n_pos=4, r=5.0) + torch.tensor([0.0, -50.0, 20.0])])
        videos.append({"image": [f.to(dev) for f in frames], "target_hw": (128, 256), "query_points": q})
    pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f32", max_batch=4, max_decode_batch=8).to(dev))
    model = SamPt(PipsPointTracker(state_dict=pips_sd), pred, sam_iou_threshold=0.1, positive_points_per_mask=4,
                  negative_points_per_mask=0, iterative_refinement_iterations=3).eval()
    graph = pred.use_graph
    pred.use_graph = False                             # plain launches: the yardstick for the replayed chains below
    ref = [model(v) for v in videos]
    torch.cuda.synchronize()
    pred.use_graph = graph
    for rep in range(3):                               # (first sight of a signature is eager, then capture, then replays)
        again = [model(v) for v in videos]
        torch.cuda.synchronize()
        for a, b in zip(again, ref):
            # replayed hipGraphs over several clips whose items switch refinement off at different passes: bitwise the eager
            # chain (a memset NODE in the captured chain used to land late on replay and re-activate items: round 3)
            assert torch.equal(torch.stack(a["logits"]), torch.stack(b["logits"])) and a["scores_per_frame"] == b["scores_per_frame"]
    assert not graph or pred.graph_stats()[2] > 0
    for rep in range(2):
        got = list(model.stream(videos))
        torch.cuda.synchronize()
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert torch.equal(a["trajectories"], b["trajectories"]) and torch.equal(a["visibilities"], b["visibilities"])
            assert a["scores_per_frame"] == b["scores_per_frame"] and a["scores"] == b["scores"]
            for la, lb in zip(a["logits"], b["logits"]):
                assert torch.equal(la, lb)
    h = model.forward_begin(videos[0])                 # a handle may be collected late, and more than once
    torch.cuda.synchronize()
    out = model.forward_end(h)
    assert model.forward_end(h) is out
    assert torch.equal(torch.stack(out["logits"]), torch.stack(ref[0]["logits"]))


# ------------------------------------------------------------------------------------------ one clip over N ranks
@pytest.mark.parametrize("tracker", ["pips", "cotracker"])
def test_frame_sharded_emulation_equals_full_forward(dev, pips_sd, tracker):
    """dist.sharded_forward's in-clip split, one GPU standing in for each rank of a 3-rank job in turn (emulate=(r, N): its
    share of the tracker encoder + the stubbed pyramid all_gather, the replicated window chain, its frame batches of the SAM
    stage): trajectories are those of the full forward bit for bit on every "rank", and the ranks' masks tile the full result."""
    from sam_pt_amd.dist import frame_batches, index_masks, sharded_forward
    from sam_pt_amd.point_tracker import CoTrackerPointTracker, PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS
    cfg = SAM_CONFIGS["vit_test"]
    frames, centres = synthetic_clip(T=10, H=128, W=256, seed=4)
    q = disc_queries(centres, n_pos=4, r=9.0)[None]
    kw = dict(sam_iou_threshold=-1e9, positive_points_per_mask=4, negative_points_per_mask=0, iterative_refinement_iterations=2)
    trk = PipsPointTracker(state_dict=pips_sd, fnet_chunk=4) if tracker == "pips" else CoTrackerPointTracker(seed=72, fnet_chunk=4)
    model = SamPt(trk, SamPredictor(SamHip(config=cfg, seed=72, precision="f32", max_batch=4).to(dev)), **kw).eval()
    video = {"image": [f.to(dev) for f in frames], "target_hw": (128, 256), "query_points": q}
    full = model(video)
    want = index_masks(torch.stack(full["logits"], dim=0))
    N = 3
    for shard_fnet in (True, False):          # the opt-in pyramid exchange, and the default (north_star) mode: mask gather only
        seen = torch.zeros(10, dtype=torch.bool)
        for r in range(N):
            masks, out = sharded_forward(model, video, batch=2, shard_fnet=shard_fnet, emulate=(r, N))
            ids = [t for b in frame_batches(10, N, r, 2) for t in b]
            assert torch.equal(out["trajectories"], full["trajectories"]) and torch.equal(out["visibilities"], full["visibilities"])
            assert torch.equal(masks, want[torch.as_tensor(ids, device=want.device)])
            if shard_fnet:
                fs = out["fnet_shard"]
                assert fs.bytes_received > 0 and fs.stub_ms() > 0.0
            else:
                assert "fnet_shard" not in out
            seen[ids] = True
        assert bool(seen.all())
