"""Drop-in proof on the CPU: the reference's OWN config tree
(``configs/model/sam_pt.yaml`` + ``point_tracker/*.yaml`` + ``sam/*.yaml``) composed and instantiated with exactly the
overrides INTEGRATION.md lists builds our seam classes; and — where /root/reference exists — the UNCHANGED reference
``SamPt`` (sam_pt/modeling/sam_pt.py), ``demo.demo.run_inference`` (demo/demo.py:114-155) and
``SamPtEvaluator.evaluate_video`` (sam_pt/vos_eval/evaluator.py:47-60) run over those classes, whose C-ABI layer is a
recording fake computing with the CPU oracle (tests/fake_hip.py), and give the oracle-driven reference protocol's results.
Hydra / OmegaConf are not installed: tests/hydra_lite.py restates the subset the model configs use."""
import os

import numpy as np
import pytest
import torch

from tests import hydra_lite as H
from tests.util import disc_queries, iou, synthetic_clip

REF = "/root/reference"
CONFIGS = os.path.join(REF, "configs")
needs_ref = pytest.mark.skipif(not os.path.isdir(CONFIGS), reason="reference tree not present (GPU box)")

# INTEGRATION.md §1, verbatim (Hydra CLI syntax)
OVERRIDES = ["model.point_tracker._target_=sam_pt_amd.point_tracker.{TRACKER}",
             "model.sam_predictor._target_=sam_pt_amd.sam_predictor.SamPredictor",
             "model.sam_predictor.sam_model._target_=sam_pt_amd.sam_predictor.SamHip",
             "+model.sam_predictor.sam_model._recursive_=false"]
TEST_GEOMETRY = ["model.sam_predictor.sam_model.image_encoder.depth=2", "model.sam_predictor.sam_model.image_encoder.embed_dim=64",
                 "model.sam_predictor.sam_model.image_encoder.num_heads=2", "model.sam_predictor.sam_model.image_encoder.window_size=6",
                 "model.sam_predictor.sam_model.image_encoder.global_attn_indexes=[1]", "model.sam_predictor.sam_model.image_size=256",
                 "model.sam_predictor.sam_model.image_embedding_size=16"]


def _compose(tracker_option, sam_option, tracker_cls, extra=()):
    cfg = {"model": H.compose(CONFIGS, "model", "sam_pt", {"point_tracker": tracker_option,
                                                           "sam@sam_predictor.sam_model": sam_option})}
    ov = [o.replace("{TRACKER}", tracker_cls) for o in OVERRIDES] + list(extra)
    H.apply_overrides(cfg, ov)
    H.apply_overrides(cfg, ["model.point_tracker.checkpoint_path=null", "model.sam_predictor.sam_model.checkpoint=null"])
    return H.resolve(cfg, cwd="/nonexistent")["model"]


@needs_ref
@pytest.mark.parametrize("tracker_option,tracker_cls,sam_option,hq", [
    ("cotracker", "CoTrackerPointTracker", "samhq_vit_huge", True),        # the shipped defaults (sam_pt.yaml:3-5)
    ("pips", "PipsPointTracker", "sam_vit_huge", False),                   # the metric's configuration
    ("pips_plus_plus", "PipsPlusPlusPointTracker", "sam_vit_base", False)])
def test_reference_yaml_instantiates_our_classes(tracker_option, tracker_cls, sam_option, hq):
    """Every keyword of the reference YAMLs is accepted; the SAM geometry comes from the nested image_encoder node."""
    import sam_pt_amd.point_tracker as P
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    cfg = _compose(tracker_option, sam_option, tracker_cls, ["model._target_=sam_pt_amd.sam_pt.SamPt"])
    model = H.instantiate(cfg)
    assert type(model) is SamPt and type(model.point_tracker) is getattr(P, tracker_cls)
    assert type(model.sam_predictor) is SamPredictor and type(model.sam_predictor.model) is SamHip
    sam = model.sam_predictor.model
    want = {"samhq_vit_huge": (1280, 32, 16, (7, 15, 23, 31)), "sam_vit_huge": (1280, 32, 16, (7, 15, 23, 31)),
            "sam_vit_base": (768, 12, 12, (2, 5, 8, 11))}[sam_option]
    assert (sam.cfg.embed_dim, sam.cfg.depth, sam.cfg.num_heads, sam.cfg.global_attn_indexes) == want
    assert sam.hq is hq and sam.image_size == 1024 and sam.image_embedding_size == 64 and sam.prompt_embed_dim == 256
    assert model.sam_iou_threshold == 0.7 and model.iterative_refinement_iterations == 12
    assert model.positive_points_per_mask == 16 and model.negative_points_per_mask == 1
    if tracker_option == "cotracker":
        t = model.point_tracker
        assert (t.interp_shape, t.visibility_threshold, t.support_grid_size, t.support_grid_every_n_frames) == ((384, 512), 0.7, 2, 12)
    if tracker_option == "pips":
        t = model.point_tracker
        assert (t.stride, t.s, t.initial_next_frame_visibility_threshold) == (4, 8, 0.9)
    model.eval()
    assert model.device.type == "cpu"                                   # .to("cuda") is what demo.load_model adds (demo.py:111)


def _small_setup(monkeypatch, neg):
    """Reference config tree at the reduced test geometry + PIPS, our seam classes over the recording fake device."""
    from tests import fake_hip
    from sam_pt_amd.weights import SAM_CONFIGS, init_pips_state_dict, init_sam_state_dict
    tcfg = SAM_CONFIGS["vit_test"]
    sd, psd = init_sam_state_dict(tcfg, 72), init_pips_state_dict(72)
    fake = fake_hip.install(monkeypatch, sd, tcfg, psd)
    cfg = _compose("pips", "sam_vit_base", "PipsPointTracker", TEST_GEOMETRY + [
        "model.iterative_refinement_iterations=2", "model.positive_points_per_mask=4", f"model.negative_points_per_mask={neg}",
        "model.sam_iou_threshold=-1.0e+9", "+model.point_tracker.state_dict=null", "+model.sam_predictor.sam_model.precision=f32"])
    cfg["point_tracker"]["state_dict"] = psd                             # no checkpoint files here: hand the weights over
    cfg["sam_predictor"]["sam_model"]["state_dict"] = sd
    frames, centres = synthetic_clip(T=10, H=128, W=256, seed=72)
    q = disc_queries(centres, n_pos=4 + neg, r=9.0)
    if neg:
        q[4:, 1:] += torch.tensor([40.0, 30.0])
    q2 = q.clone()
    q2[:, 1:] += torch.tensor([-60.0, 25.0])
    qp = torch.stack([q, q2])
    return fake, cfg, tcfg, sd, psd, frames, qp


def _oracle_reference(tcfg, sd, psd, frames, qp, neg):
    from oracle.parity import reference_run
    kw = dict(sam_iou_threshold=-1e9, positive_points_per_mask=4, negative_points_per_mask=neg,
              iterative_refinement_iterations=2, point_tracker_mask_batch_size=5)
    return reference_run(tcfg, sd, psd, frames, qp, kw)


@needs_ref
@pytest.mark.parametrize("neg", [0, 1])
def test_unchanged_reference_sampt_runs_over_our_seam_classes(monkeypatch, neg):
    """The reference's own SamPt class, built from its own YAML with only the two seam `_target_`s swapped, drives our
    PipsPointTracker / SamPredictor through set_image / predict_torch / evaluate_batch and reproduces the oracle protocol."""
    from oracle import reference_loader as RL
    RefSamPt = RL.load_sam_pt()
    fake, cfg, tcfg, sd, psd, frames, qp = _small_setup(monkeypatch, neg)
    assert cfg["_target_"] == "sam_pt.modeling.sam_pt.SamPt"             # untouched: the reference class itself
    model = H.instantiate(cfg).eval()
    assert type(model) is RefSamPt
    video = {"image": [f for f in frames], "target_hw": (128, 256), "query_points": qp}
    out = model(video)
    ref = _oracle_reference(tcfg, sd, psd, frames, qp, neg)
    assert torch.equal(out["visibilities"], ref["visibilities"])
    assert (out["trajectories"].round() == ref["trajectories"].round()).all()
    assert (out["trajectories"] - ref["trajectories"]).abs().max() < 1e-3
    for m in range(2):
        a, b = out["logits"][m], ref["logits"][m]
        assert (a - b).abs().max() < 1e-4
        assert min(iou(a[t] > 0, b[t] > 0) for t in range(10)) >= 1 - 1e-3
    assert fake.calls["vit_encode_frames"] == 12 and fake.calls["sam_decode"] == 2 * (1 + neg + 2) + 20 * (1 + (1 if neg else 0) + 2)
    assert fake.calls["pips_fnet_frames"] == 10                         # every frame encoded once for both objects' points


@needs_ref
def test_reference_demo_and_evaluator_entry_points(monkeypatch):
    """demo.demo.run_inference and SamPtEvaluator.evaluate_video (both call `model(video)` and unpack the result dict)
    over our device-path SamPt built from the reference YAML with `model._target_` swapped too."""
    from oracle import reference_loader as RL
    from sam_pt_amd.sam_pt import SamPt
    demo = RL.load_demo()
    Evaluator = RL.load_evaluator()
    fake, cfg, tcfg, sd, psd, frames, qp = _small_setup(monkeypatch, 0)
    cfg["_target_"] = "sam_pt_amd.sam_pt.SamPt"
    model = H.instantiate(cfg).eval()
    assert type(model) is SamPt
    logits, traj, vis, scores = demo.run_inference(model, frames, qp, (128, 256))
    ref = _oracle_reference(tcfg, sd, psd, frames, qp, 0)
    assert logits.shape == (10, 3, 128, 256) and (logits[:, 0] == 0).all()
    assert torch.equal(vis, ref["visibilities"]) and (traj.round() == ref["trajectories"].round()).all()
    for m in range(2):
        assert min(iou(logits[t, m + 1] > 0, ref["logits"][m][t] > 0) for t in range(10)) >= 1 - 1e-3
    assert fake.calls["sam_track_decode_items"] == 20 and fake.calls["sam_decode"] == 0      # fused device protocol
    ev = Evaluator(cfg=None, model=model)
    out = ev.evaluate_video({"video_name": "synthetic", "video_id": 0, "image": [f for f in frames], "target_hw": (128, 256),
                             "query_points": qp})
    assert set(out) == {"logits", "trajectories", "visibilities", "scores"}
    assert torch.equal(out["visibilities"], vis) and len(out["logits"]) == 2 and len(out["scores"]) == 2


def test_hydra_lite_interpolation_and_instantiate(tmp_path):
    """The stand-in itself: defaults composition, package placement, override, relative interpolation, _partial_."""
    (tmp_path / "g" / "sub").mkdir(parents=True)
    (tmp_path / "g" / "base.yaml").write_text("defaults:\n  - sub: a\nsize: 4\nsub:\n  w: ${ ..size }\n")
    (tmp_path / "g" / "top.yaml").write_text("defaults:\n  - base\n  - override sub: b\n  - _self_\nsize: 8\n"
                                             "f:\n  _target_: builtins.int\n  _partial_: true\n  base: 2\n")
    (tmp_path / "g" / "sub" / "a.yaml").write_text("name: a\n")
    (tmp_path / "g" / "sub" / "b.yaml").write_text("name: b\nroot: ${hydra:runtime.cwd}/x\n")
    cfg = H.resolve(H.compose(str(tmp_path), "g", "top"), cwd="/cwd")
    assert cfg["size"] == 8 and cfg["sub"] == {"name": "b", "root": "/cwd/x", "w": 8}
    assert H.instantiate(cfg)["f"]("101") == 5


def test_prepared_pyramid_path_on_the_fake_device(monkeypatch):
    """PipsPointTracker.prepare() + forward() (the split SamPt's stream schedule uses: pyramid first, window rounds later)
    gives the plain forward()'s result, reuses the pyramid (every frame encoded once) and drops a stale cache entry."""
    from tests import fake_hip
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.weights import init_pips_state_dict
    psd = init_pips_state_dict(72)
    fake = fake_hip.install(monkeypatch, pips_sd=psd)
    frames, centres = synthetic_clip(T=9, H=128, W=256, seed=72)
    q = torch.cat([disc_queries(centres, n_pos=3, r=9.0, t=0), disc_queries(centres, n_pos=1, r=4.0, t=6)])[None]
    trk = PipsPointTracker(state_dict=psd, fnet_chunk=4)
    tr0, vi0 = trk(frames[None], q)
    assert fake.calls["pips_fnet_frames"] == 9
    rgbs = frames[None].clone()
    trk.prepare(rgbs[0])
    tr1, vi1 = trk(rgbs, q)
    assert fake.calls["pips_fnet_frames"] == 18 and torch.equal(tr0, tr1) and torch.equal(vi0, vi1)
    rgbs[0, 0, 0, 0, 0] += 1                                         # in-place edit: the cached pyramid is stale
    trk(rgbs, q)
    assert fake.calls["pips_fnet_frames"] == 27


def test_stream_of_clips_on_the_fake_device(monkeypatch):
    """SamPt.stream / forward_begin / forward_end (a clip in flight ahead of the one being collected) on the recording fake:
    clip by clip the result of forward(); on a device without streams the handle is simply complete."""
    from tests import fake_hip
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import PendingForward, SamPt
    from sam_pt_amd.weights import SAM_CONFIGS, init_pips_state_dict, init_sam_state_dict
    tcfg = SAM_CONFIGS["vit_test"]
    sd, psd = init_sam_state_dict(tcfg, 72), init_pips_state_dict(72)
    fake_hip.install(monkeypatch, sd, tcfg, psd)
    videos = []
    for seed, T in ((72, 6), (73, 4)):
        frames, centres = synthetic_clip(T=T, H=128, W=256, seed=seed)
        videos.append({"image": [f for f in frames], "target_hw": (128, 256), "query_points": disc_queries(centres, n_pos=4, r=9.0)[None]})
    model = SamPt(PipsPointTracker(state_dict=psd), SamPredictor(SamHip(config=tcfg, state_dict=sd, precision="f32")),
                  sam_iou_threshold=-1e9, positive_points_per_mask=4, negative_points_per_mask=0,
                  iterative_refinement_iterations=1).eval()
    ref = [model(v) for v in videos]
    got = list(model.stream(videos))
    assert len(got) == 2
    for a, b in zip(got, ref):
        assert torch.equal(a["trajectories"], b["trajectories"]) and a["scores_per_frame"] == b["scores_per_frame"]
        assert torch.equal(torch.stack(a["logits"]), torch.stack(b["logits"]))
    h = model.forward_begin(videos[1])
    assert isinstance(h, PendingForward) and model.forward_end(h) is model.forward_end(h)
    assert list(model.stream([])) == []


def test_dead_row_cache_host_logic_on_the_fake_device(monkeypatch):
    """SamPredictor.encode_frames around sampt_vit_encode_live: one cache per frame geometry, built once from the first frame
    of that size and reused by later clips; square frames (nothing to skip) take the plain entry point; the embeddings equal
    the plain path's (the fake computes both with the oracle encoder)."""
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    from tests import fake_hip
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    fake = fake_hip.install(monkeypatch, sd, cfg, None)
    fake.model_live_rows = True
    pred = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32"))
    land, _ = synthetic_clip(T=3, H=144, W=256, seed=5)
    a = pred.encode_frames(land)                                   # 9 token rows -> 12 live rows of 16
    assert fake.calls["vit_dead_cache_builds"] == 1 and fake.calls["vit_encode_live_frames"] == 3
    pred.encode_frames(land[:2])                                   # second clip, same geometry: cache reused
    assert fake.calls["vit_dead_cache_builds"] == 1 and fake.calls["vit_encode_live_frames"] == 5
    low, _ = synthetic_clip(T=1, H=80, W=256, seed=6)
    pred.encode_frames(low)                                        # another geometry: its own cache
    assert fake.calls["vit_dead_cache_builds"] == 2
    sq, _ = synthetic_clip(T=1, H=256, W=256, seed=7)
    n_plain = fake.calls["vit_encode_frames"]
    pred.encode_frames(sq)                                         # square: nothing to skip, plain entry point
    assert fake.calls["vit_dead_cache_builds"] == 2 and fake.calls["vit_encode_live_frames"] == 6
    assert fake.calls["vit_encode_frames"] == n_plain + 1
    pred.skip_dead_rows = False
    b = pred.encode_frames(land)
    assert fake.calls["vit_encode_live_frames"] == 6 and torch.equal(a, b)


def test_checkpoint_files_load_through_the_reference_conventions(tmp_path, monkeypatch):
    """`checkpoint` / `checkpoint_path` constructor arguments: SAM `.pth` state dict (sam.py:21-24), PIPS directory with
    `model-*.pth` holding 'model_state_dict' (utils/saverloader.py:30-73), CoTracker `.pth` optionally wrapped in 'model'
    (build_cotracker) — the weights that reach the packer are the file's, and HQ-SAM is recognised by its keys."""
    from tests import fake_hip
    from sam_pt_amd.point_tracker import CoTrackerPointTracker, PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.weights import SAM_CONFIGS, init_cotracker_state_dict, init_pips_state_dict, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 5, hq=True)
    torch.save(sd, tmp_path / "sam_hq_vit_test.pth")
    sam = SamHip(config=cfg, checkpoint=str(tmp_path / "sam_hq_vit_test.pth"), precision="f32")
    assert sam.hq and all(torch.equal(sam.sd[k], v) for k, v in sd.items())
    plain = {k: v for k, v in sd.items() if not any(t in k for t in ("hf_token", "hf_mlp", "compress_vit_feat",
                                                                        "embedding_encoder", "embedding_maskfeature"))}
    torch.save(plain, tmp_path / "sam_vit_test.pth")
    assert not SamHip(config=cfg, checkpoint=str(tmp_path / "sam_vit_test.pth")).hq
    with pytest.raises(ValueError):
        SamHip(config=cfg, checkpoint=str(tmp_path / "sam_vit_test.pth"), hq=True)
    psd = init_pips_state_dict(9)
    (tmp_path / "pips").mkdir()
    torch.save({"model_state_dict": init_pips_state_dict(8)}, tmp_path / "pips" / "model-000000001.pth")
    torch.save({"model_state_dict": psd}, tmp_path / "pips" / "model-000200000.pth")          # the newest one wins
    trk = PipsPointTracker(checkpoint_path=str(tmp_path / "pips"))
    assert all(torch.equal(trk._sd[k], v) for k, v in psd.items())
    csd = init_cotracker_state_dict(3)
    torch.save({"model": csd}, tmp_path / "cotracker_stride_4_wind_8.pth")
    ctrk = CoTrackerPointTracker(checkpoint_path=str(tmp_path / "cotracker_stride_4_wind_8.pth"))
    assert all(torch.equal(ctrk._sd[k], v) for k, v in csd.items())
    # ... and they reach the device-side packer unchanged in value (fake device: the packed dict is what *_create receives)
    fake = fake_hip.install(monkeypatch, plain, cfg, psd)
    pred = SamPredictor(SamHip(config=cfg, checkpoint=str(tmp_path / "sam_vit_test.pth"), precision="f32"))
    pred._ensure()
    assert torch.equal(pred._wv["image_encoder.blocks.0.attn.qkv.weight"], plain["image_encoder.blocks.0.attn.qkv.weight"])
    assert torch.equal(pred._wd["mask_decoder.iou_prediction_head.layers.2.weight"], plain["mask_decoder.iou_prediction_head.layers.2.weight"])
