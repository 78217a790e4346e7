#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the REFERENCE ITSELF (run in the build container only).

TEST INFRASTRUCTURE.  The reference has no tests and no golden vectors, so the pins are outputs of the
reference's own code run in place from /root/reference (never copied):

  pips_window.npz   sam_pt.point_tracker.pips.pips.Pips.forward           (one 8-frame window, 6 iterations)
  pips_tracker.npz  sam_pt.point_tracker.pips.tracker.PipsPointTracker    (T=12 clip, queries at t=0, 5, 11)
  sampt_ref.npz     sam_pt.modeling.sam_pt.SamPt.forward                  (reference orchestration; predictor = the
                                                                           SAM oracle, tracker = reference PIPS)
  sam_hf.npz        HuggingFace transformers SamModel                     (secondary pin of the absent third-party SAM)
  sampt_patch.npz   SamPt.forward with use_patch_matching_filtering (rgb2lab substituted by our restatement)
  pips2.npz         sam_pt.point_tracker.pips_plus_plus.{PipsPlusPlus, PipsPlusPlusPointTracker}   (row f4)
  sam_hq_hf.npz     HuggingFace transformers SamHQModel                   (secondary pin of the absent HQ-SAM decoder)

Weights are NOT stored: they are regenerated from the seed (sam_pt_amd/weights.py), inputs from
sam_pt_amd/synth.py.  Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transformers  # noqa: E402,F401  (import before the reference's third-party stubs are registered)
from oracle import hf_crosscheck as H  # noqa: E402
from oracle import reference_loader as RL  # noqa: E402
from oracle import sam_ref as R  # noqa: E402
from sam_pt_amd.synth import disc_queries, synthetic_clip  # noqa: E402
from sam_pt_amd.weights import SAM_CONFIGS, init_pips_state_dict, init_sam_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def golden_queries(centres):
    return torch.cat([disc_queries(centres, n_pos=4, r=9.0, t=0), disc_queries(centres, n_pos=2, r=6.0, t=5),
                      disc_queries(centres, n_pos=1, r=3.0, t=11)])[None]


def sampt_kwargs(npos, neg):
    return dict(sam_iou_threshold=0.06, positive_points_per_mask=npos, negative_points_per_mask=neg,
                iterative_refinement_iterations=3, point_tracker_mask_batch_size=5,
                positive_point_selection_method="kmedoids", negative_point_selection_method="mixed",
                add_other_objects_positive_points_as_negative_points=True, max_other_objects_positive_points=None,
                use_patch_matching_filtering=False, patch_size=3, patch_similarity_threshold=0.01, use_point_reinit=False,
                reinit_point_tracker_horizon=24, reinit_horizon=24, reinit_variant="reinit-at-median-of-area-diff")


def sampt_video(frames, centres, npos, neg):
    q = disc_queries(centres, n_pos=npos + neg, r=9.0)
    if neg:
        q[npos:, 1:] += torch.tensor([40.0, 30.0])
    q2 = disc_queries(centres, n_pos=npos + neg, r=5.0)
    q2[:, 1:] += torch.tensor([-60.0, 20.0])
    return {"image": [f for f in frames], "target_hw": tuple(frames.shape[-2:]), "query_points": torch.stack([q, q2])}


def reinit_kwargs(variant, neg):
    kw = sampt_kwargs(4, neg)
    kw.update(use_point_reinit=True, reinit_horizon=5, reinit_point_tracker_horizon=6, reinit_variant=variant,
              positive_point_selection_method="random", negative_point_selection_method="random", sam_iou_threshold=-1e9)
    return kw


def reinit_video(frames, centres, neg):
    video = sampt_video(frames, centres, 4, neg)
    video["query_points"][1, :, 0] = 2           # the second object starts later
    return video


def query_mask_video(frames, centres):
    H, W = frames.shape[-2:]
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    m0 = (((xx - centres[0, 0]) ** 2 + (yy - centres[0, 1]) ** 2) <= 15 ** 2).float()
    m1 = ((xx > 150) & (xx < 190) & (yy > 20) & (yy < 50)).float()
    return {"image": [f for f in frames], "target_hw": (H, W), "query_masks": torch.stack([m0, m1]),
            "query_point_timestep": torch.tensor([0.0, 3.0])}


def make_hq_golden():
    """HF SamHQModel on the reduced geometry (secondary pin of the absent third-party HQ-SAM decoder)."""
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72, hq=True)
    hf = H.build_hf_hq_model(cfg, sd)
    x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(3))
    emb, interm = H.hf_hq_embed(hf, x)
    pts = torch.tensor([[[30.5, 40.2], [100.0, 200.0], [220.0, 15.0]]])
    lab = torch.tensor([[1, 0, 1]])
    low, iou = H.hf_hq_decode(hf, emb, interm, pts, lab)
    box = torch.tensor([[10.0, 20.0, 200.0, 180.0]])
    low2, iou2 = H.hf_hq_decode(hf, emb, interm, pts, lab, boxes=box, masks=low)
    np.savez_compressed(os.path.join(OUT, "sam_hq_hf.npz"), emb=emb.numpy(), interm=interm[0].numpy()[:, ::4, ::4],
                        pts=pts.numpy(), lab=lab.numpy(), low=low.numpy(), iou=iou.numpy(), box=box.numpy(),
                        low2=low2.numpy(), iou2=iou2.numpy())


def patch_kwargs():
    kw = sampt_kwargs(4, 1)
    kw.update(use_patch_matching_filtering=True, patch_size=3, patch_similarity_threshold=0.6,
              positive_point_selection_method="random", negative_point_selection_method="random", sam_iou_threshold=-1e9)
    return kw


def make_patch_golden():
    """Reference SamPt with the patch-similarity filter on (sam_pt.py:597-682).  skimage is absent, so the reference's
    ``color.rgb2lab`` is substituted by our restatement: this pins everything of the filter except that one function."""
    import sys as _sys
    from sam_pt_amd.sam_pt import rgb2lab
    RefSamPt = RL.load_sam_pt()
    _sys.modules["skimage"].color.rgb2lab = lambda a: rgb2lab(torch.from_numpy(np.ascontiguousarray(a))).numpy()
    mod = _sys.modules[RefSamPt.__module__]
    if hasattr(mod, "color"):
        mod.color.rgb2lab = _sys.modules["skimage"].color.rgb2lab
    _, PipsPointTracker, _ = RL.load_pips()
    psd = init_pips_state_dict(72)
    frames, centres = synthetic_clip(T=12, H=128, W=256, seed=72)
    d = tempfile.mkdtemp()
    torch.save({"model_state_dict": psd}, os.path.join(d, "model-000000001.pth"))
    trk = PipsPointTracker(checkpoint_path=d, stride=4, s=8).eval()
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    pred = R.SamPredictorRef(sd, cfg)
    pred.model = torch.nn.Module()
    pred.model.device, pred.model.mask_threshold = torch.device("cpu"), 0.0
    video = sampt_video(frames[:10], centres, 4, 1)
    video["query_points"][1, :, 0] = 3                                     # second object: queries in the middle
    res = RefSamPt(trk, pred, **patch_kwargs()).eval()(video)
    masks = torch.stack([l > 0 for l in res["logits"]])
    np.savez_compressed(os.path.join(OUT, "sampt_patch.npz"), masks=np.packbits(masks.numpy(), axis=-1),
                        traj=res["trajectories"].numpy(), vis=res["visibilities"].numpy(),
                        scores_per_frame=np.array(res["scores_per_frame"], dtype=np.float32))
    print("visibility codes in the golden:", sorted(set(res["visibilities"].flatten().tolist())))


def make_pips2_golden():
    """Reference PipsPlusPlus / PipsPlusPlusPointTracker (run in place) on the synthetic clip."""
    from sam_pt_amd.weights import init_pips2_state_dict
    Pips2, Tracker2 = RL.load_pips2()
    sd = init_pips2_state_dict(72)
    frames, centres = synthetic_clip(T=12, H=128, W=256, seed=72)
    m = Pips2(stride=8).eval()
    m.load_state_dict(sd, strict=True)
    q = disc_queries(centres, n_pos=5, r=9.0)
    out = {}
    with torch.no_grad():
        preds, _, feats, _ = m(q[None, :, 1:].repeat(12, 1, 1)[None], frames[None].float(), iters=6)
        fm = m.fnet(2 * (frames.float() / 255.0) - 1.0)
    out["model_xys"] = q[:, 1:].numpy()
    out["model_traj_iter0"], out["model_traj"] = preds[0][0].numpy(), preds[-1][0].numpy()
    out["model_feats2"] = feats[1][0].numpy()
    out["fmap_patch"] = fm[:, :, 4:8, 12:16].numpy()
    out["fmap_abs_mean"] = fm.abs().mean(dim=(1, 2, 3)).numpy()
    d = tempfile.mkdtemp()
    torch.save({"model_state_dict": sd}, os.path.join(d, "model-000000001.pth"))
    for name, t, maxlen, iters in (("t0", 0, 128, 16), ("t5", 5, 128, 8), ("t5_chunked", 5, 5, 3)):
        trk = Tracker2(checkpoint_path=d, stride=8, max_sequence_length=maxlen, iters=iters, image_size=None).eval()
        qq = disc_queries(centres, n_pos=4, r=9.0, t=t)[None]
        with torch.no_grad():
            tr, vi = trk(frames[None].float(), qq.clone())
        out[f"trk_{name}_q"], out[f"trk_{name}_traj"], out[f"trk_{name}_vis"] = qq.numpy(), tr.numpy(), vi.numpy()
    trk = Tracker2(checkpoint_path=d, stride=8, max_sequence_length=128, iters=4, image_size=(128, 192)).eval()
    qq = disc_queries(centres, n_pos=4, r=9.0, t=3)[None]
    with torch.no_grad():
        tr, _ = trk(frames[None].float(), qq.clone())
    out["trk_resized_q"], out["trk_resized_traj"] = qq.numpy(), tr.numpy()
    np.savez_compressed(os.path.join(OUT, "pips2.npz"), **out)


def main():
    assert RL.available(), "the reference tree is required to (re)generate goldens"
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    Pips, PipsPointTracker, _ = RL.load_pips()
    psd = init_pips_state_dict(72)
    frames, centres = synthetic_clip(T=12, H=128, W=256, seed=72)

    # ---- one window through the reference Pips
    m = Pips(S=8, stride=4).eval()
    m.load_state_dict(psd, strict=True)
    q = disc_queries(centres, n_pos=5, r=9.0)
    with torch.no_grad():
        preds, _, vis, ffeat, _ = m(q[None, :, 1:], frames[None, :8].float(), iters=6, return_feat=True)
        fm = m.fnet(2 * (frames[:8].float() / 255.0) - 1.0)
    np.savez_compressed(os.path.join(OUT, "pips_window.npz"), xys=q[:, 1:].numpy(), traj=preds[-1][0].numpy(),
                        traj_iter0=preds[0][0].numpy(), vis_logits=vis[0].numpy(), ffeat=ffeat[0].numpy(),
                        fmap_patch=fm[:, :, 8:16, 24:32].numpy(), fmap_abs_mean=fm.abs().mean(dim=(1, 2, 3)).numpy())

    # ---- the whole reference tracker
    d = tempfile.mkdtemp()
    torch.save({"model_state_dict": psd}, os.path.join(d, "model-000000001.pth"))
    trk = PipsPointTracker(checkpoint_path=d, stride=4, s=8).eval()
    gq = golden_queries(centres)
    with torch.no_grad():
        tr, vi = trk(frames[None], gq)
    np.savez_compressed(os.path.join(OUT, "pips_tracker.npz"), query_points=gq.numpy(), traj=tr.numpy(), vis=vi.numpy())

    # ---- the reference SamPt orchestration (predictor = SAM oracle on CPU)
    RefSamPt = RL.load_sam_pt()
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    out = {}
    for neg in (0, 2):
        pred = R.SamPredictorRef(sd, cfg)
        pred.model = torch.nn.Module()
        pred.model.device, pred.model.mask_threshold = torch.device("cpu"), 0.0
        video = sampt_video(frames[:10], centres, 4, neg)
        res = RefSamPt(trk, pred, **sampt_kwargs(4, neg)).eval()(video)
        masks = torch.stack([l > 0 for l in res["logits"]])                       # (M,T,H,W) bool
        out[f"masks_neg{neg}"] = np.packbits(masks.numpy(), axis=-1)
        out[f"finite_neg{neg}"] = torch.stack([torch.isfinite(l).all(-1).all(-1) for l in res["logits"]]).numpy()
        out[f"scores_per_frame_neg{neg}"] = np.array(res["scores_per_frame"], dtype=np.float32)
        out[f"traj_neg{neg}"] = res["trajectories"].numpy()
        out[f"vis_neg{neg}"] = res["visibilities"].numpy()
        out[f"predict_calls_neg{neg}"] = np.array([pred.n_set_image, pred.n_predict])
    np.savez_compressed(os.path.join(OUT, "sampt_ref.npz"), **out)

    # ---- reference SamPt with point re-initialisation and in query_masks mode (random point selection, seeded RNG)
    out = {}
    for name, kw, video in [("reinit_median", reinit_kwargs("reinit-at-median-of-area-diff", 0), reinit_video(frames, centres, 0)),
                            ("reinit_sync", reinit_kwargs("reinit-on-similar-mask-area-and-sync-masks", 1), reinit_video(frames, centres, 1)),
                            ("qmasks", dict(sampt_kwargs(4, 1), positive_point_selection_method="random",
                                            negative_point_selection_method="random", sam_iou_threshold=-1e9),
                             query_mask_video(frames[:8], centres))]:
        pred = R.SamPredictorRef(sd, cfg)
        pred.model = torch.nn.Module()
        pred.model.device, pred.model.mask_threshold = torch.device("cpu"), 0.0
        torch.manual_seed(5)
        res = RefSamPt(trk, pred, **kw).eval()(video)
        masks = torch.stack([l > 0 for l in res["logits"]])
        out[f"{name}_masks"] = np.packbits(masks.numpy(), axis=-1)
        out[f"{name}_traj"] = res["trajectories"].numpy()
        out[f"{name}_vis"] = res["visibilities"].numpy()
        out[f"{name}_scores_per_frame"] = np.array(res["scores_per_frame"], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "sampt_reinit.npz"), **out)

    # ---- HF SamModel on the reduced geometry
    hf = H.build_hf_model(cfg, sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 256, 256, generator=g)
    emb = H.hf_embed(hf, x)
    pts = torch.tensor([[[30.5, 40.2], [100.0, 200.0], [220.0, 15.0]]])
    lab = torch.tensor([[1, 0, 1]])
    low, iou = H.hf_decode(hf, emb, pts, lab)
    box = torch.tensor([[10.0, 20.0, 200.0, 180.0]])
    low2, iou2 = H.hf_decode(hf, emb, pts, lab, boxes=box, masks=low)
    # x is NOT stored: tests regenerate it from the same generator seed (3)
    np.savez_compressed(os.path.join(OUT, "sam_hf.npz"), emb=emb.numpy(), pts=pts.numpy(), lab=lab.numpy(),
                        low=low.numpy(), iou=iou.numpy(), box=box.numpy(), low2=low2.numpy(), iou2=iou2.numpy())
    make_hq_golden()
    make_pips2_golden()
    make_patch_golden()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
