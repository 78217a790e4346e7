"""Parity checker shared by the GPU tests, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
TEST INFRASTRUCTURE ONLY — nothing under ``sam_pt_amd/`` imports this file.

``reference_run`` drives the reference protocol on the CPU: where ``/root/reference`` exists (the build container) the
REFERENCE'S OWN ``SamPt`` class, imported in place (``oracle/reference_loader.load_sam_pt``) — so a regression in this package's
call-by-call branch cannot move both sides of a comparison; elsewhere (the GPU box) this package's ``SamPt`` host logic, pinned
bit-identical to the reference's on ``tests/golden/sampt_ref.npz`` (``test_reference_run_drivers_agree`` checks that the two
drivers give the same result) — over the oracle tracker (``oracle/pips_ref.py``, pinned on the
reference's own PIPS) and the oracle predictor (``oracle/sam_ref.py``, pinned on HF transformers) call by call —
``set_image`` per frame, 1-2 + R ``predict_torch`` per (frame, object) — i.e. sam_pt/modeling/sam_pt.py:545-576 and
:694-866 (``predict_mask`` :760-837, the frame loop :848-858).  The SAM stage may be restricted to a subset of frames
(``frame_ids``) because a ViT-H encoder pass costs seconds of CPU; the tracker always sees the whole clip.

``compare`` turns a device result + a reference result into the numbers of SURVEY.md §8d's parity gates: per-frame
mask IoU (bar: >= 1 - 1e-3), trajectories identical in index space (``round``), visibilities identical.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


def reference_run(cfg, sd, psd, frames: torch.Tensor, query_points: torch.Tensor, sampt_kwargs: dict,
                  frame_ids: Optional[Sequence[int]] = None, hq: bool = False, reference_cost: bool = False,
                  threads: Optional[int] = None, tracker_factory=None, use_reference: Optional[bool] = None) -> Dict:
    """frames uint8 (T,3,H,W) on the CPU; query_points (M,P,3).  Returns trajectories (T,M,P,2), visibilities (T,M,P)
    [after the border rule of sam_pt.py:684-690], logits (M,len(frame_ids),H,W), scores_per_frame, the oracle
    embeddings of the SAM frames and the seconds each stage took on this host.  ``tracker_factory()`` -> an oracle tracker
    with ``forward(rgbs, query_points)`` (default: the PIPS oracle on ``psd``; ``reference_cost`` makes it spend the
    reference's redundant work too — fnet per window, init pass).  ``use_reference``: drive the reference's own ``SamPt``
    class (default: whenever /root/reference is importable), False = this package's host logic."""
    from oracle import pips_ref as PO
    from oracle import sam_ref as R
    from sam_pt_amd.point_tracker import PointTracker
    from sam_pt_amd.sam_pt import SamPt
    if threads:
        torch.set_num_threads(threads)
    T = frames.shape[0]
    ids = list(range(T)) if frame_ids is None else [int(i) for i in frame_ids]
    sec = {"tracker": 0.0, "encoder": 0.0, "decoder": 0.0, "windows": 0}
    embeddings: List[torch.Tensor] = []

    class OracleTracker(PointTracker):
        def forward(self, rgbs, qp):
            t0 = time.perf_counter()
            trk = tracker_factory() if tracker_factory is not None else PO.PipsTrackerRef(psd, reference_cost=reference_cost)
            out = trk.forward(rgbs.cpu(), qp.cpu())
            sec["tracker"] += time.perf_counter() - t0
            sec["windows"] += getattr(trk, "n_windows", 0)
            return out

    class TimedPredictor(R.SamPredictorRef):
        def set_image(self, image):
            t0 = time.perf_counter()
            super().set_image(image)
            sec["encoder"] += time.perf_counter() - t0
            embeddings.append(self.features[0].clone())

        def predict_torch(self, *a, **k):
            t0 = time.perf_counter()
            out = super().predict_torch(*a, **k)
            sec["decoder"] += time.perf_counter() - t0
            return out

    pred = TimedPredictor(sd, cfg, hq=hq)
    model = _driver(OracleTracker(), pred, sampt_kwargs, use_reference)
    with torch.no_grad():
        traj, vis = model._track_points(frames, query_points)
        sel = torch.as_tensor(ids, dtype=torch.long)
        args = (frames[sel], traj[sel], vis[sel])
        # (this package's SamPt takes the clip's precomputed embeddings as a 4th argument: none here)
        _, logits, spf = model._apply_sam_to_trajectories(*args) if _is_reference(model) else model._apply_sam_to_trajectories(*args, None)
    sec["predict_calls"] = pred.n_predict
    return {"trajectories": traj, "visibilities": vis, "logits": logits, "scores_per_frame": spf, "frame_ids": ids,
            "embeddings": torch.stack(embeddings) if embeddings else None, "seconds": sec,
            # which SamPt class drove the oracle modules: recorded in the cache file and in bench.py's parity block
            "driver": "reference SamPt (/root/reference)" if _is_reference(model) else "sam_pt_amd.SamPt host logic (pinned on "
                      "tests/golden/sampt_ref.npz)"}


# constructor keywords of the reference SamPt (sam_pt/modeling/sam_pt.py:28-50 has no defaults) = configs/model/sam_pt.yaml,
# which are also the defaults of this package's SamPt
_REF_DEFAULTS = dict(positive_point_selection_method="kmedoids", negative_point_selection_method="mixed", positive_points_per_mask=8,
                     negative_points_per_mask=0, add_other_objects_positive_points_as_negative_points=True,
                     max_other_objects_positive_points=None, point_tracker_mask_batch_size=5, iterative_refinement_iterations=12,
                     use_patch_matching_filtering=False, patch_size=3, patch_similarity_threshold=0.01, use_point_reinit=False,
                     reinit_point_tracker_horizon=24, reinit_horizon=24, reinit_variant="reinit-at-median-of-area-diff")


def _is_reference(model) -> bool:
    return type(model).__module__.startswith("sam_pt.")


def _driver(tracker, predictor, sampt_kwargs: dict, use_reference: Optional[bool]):
    """The object whose ``_track_points`` / ``_apply_sam_to_trajectories`` run the protocol: the reference's own SamPt when it
    can be imported in place, this package's otherwise."""
    from oracle import reference_loader as RL
    if use_reference is None:
        use_reference = RL.available()
    if use_reference:
        RefSamPt = RL.load_sam_pt()
        if not hasattr(predictor, "model") or predictor.model is None or not isinstance(predictor.model, torch.nn.Module):
            m = torch.nn.Module()                      # the reference keeps ``sam_predictor.model`` as a sub-module (sam_pt.py:96)
            m.device, m.mask_threshold = torch.device("cpu"), getattr(getattr(predictor, "model", None), "mask_threshold", 0.0)
            predictor.model = m
        return RefSamPt(tracker, predictor, **{**_REF_DEFAULTS, **sampt_kwargs}).eval()
    from sam_pt_amd.sam_pt import SamPt
    return SamPt(tracker, predictor, **sampt_kwargs).eval()


def mask_iou(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.cpu().bool(), b.cpu().bool()
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u


BOUNDARY_PX = 2e-3        # half-width of the band around x.5 in which round() of the oracle's own coordinate is not meaningful


def compare(out: Dict, ref: Dict) -> Dict:
    """out: SamPt.forward result of the device path (logits list[M] of (T or len(frame_ids), H, W); trajectories,
    visibilities for the whole clip).  ref: ``reference_run`` result.  Logits are matched on ref["frame_ids"]: if ``out``
    holds the whole clip they index it, otherwise it must hold exactly those frames."""
    ids = ref["frame_ids"]
    tr_o, vi_o = out["trajectories"].cpu(), out["visibilities"].cpu()
    tr_r, vi_r = ref["trajectories"], ref["visibilities"]
    # "identical in index space" = equal after round().  A coordinate that sits within ``BOUNDARY_PX`` of a rounding boundary
    # (x.5) in the oracle is ambiguous for ANY implementation that is not bit-identical to it; those are counted apart:
    # ``traj_index_identical`` is the strict statement, ``traj_index_identical_off_boundary`` ignores the ambiguous ones.
    differs = tr_o.round() != tr_r.round()
    near = ((tr_r - tr_r.floor()) - 0.5).abs() < BOUNDARY_PX
    res = {"traj_index_identical": not bool(differs.any()),
           "traj_index_identical_off_boundary": not bool((differs & ~near).any()),
           "traj_index_differing": int(differs.sum()), "traj_coords_near_boundary": int(near.sum()),
           "vis_identical": bool((vi_o == vi_r).all()),
           "traj_max_abs_px": float((tr_o - tr_r).abs().max())}
    ious, finite_ok, max_logit_err = [], True, 0.0
    M = len(out["logits"])
    cached = bool(ref.get("cached"))          # compact form of oracle/cache.py: exact masks / rejections, sub-sampled logits
    for m in range(M):
        lo = out["logits"][m].cpu()
        if lo.shape[0] != len(ids):
            lo = lo[torch.as_tensor(ids, dtype=torch.long)]
        fo = torch.isfinite(lo).flatten(1).all(1)
        if cached:
            from oracle.cache import LOGIT_STRIDE
            fr, mr, lr = ref["finite"][m], ref["masks"][m], ref["logits_sub"][m]
            lo_cmp = lo[..., ::LOGIT_STRIDE, ::LOGIT_STRIDE]
        else:
            lr = ref["logits"][m]
            fr, mr, lo_cmp = torch.isfinite(lr).flatten(1).all(1), lr > 0, lo
        finite_ok &= bool((fo == fr).all())
        for j in range(len(ids)):
            ious.append(mask_iou(lo[j] > 0, mr[j]))
            if bool(fo[j]) and bool(fr[j]):
                max_logit_err = max(max_logit_err, float((lo_cmp[j] - lr[j]).abs().max()))
    res.update({"mask_iou_min": float(min(ious)), "mask_iou_mean": float(np.mean(ious)), "masks_compared": len(ious),
                "rejections_identical": finite_ok, "logit_max_abs": max_logit_err, "frames_checked": list(ids),
                "oracle": "cached run (oracle/cache.py: exact masks, logits on every 8th pixel)" if cached else "live run"})
    return res
