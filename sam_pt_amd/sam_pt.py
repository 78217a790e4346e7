"""``SamPt`` — host orchestration of SAM-PT with the same constructor and ``forward(video)`` contract as the reference
(sam_pt/modeling/sam_pt.py:21-236), re-hosted for a device-resident pipeline.

The reference drives its predictor one call at a time with a host round trip after every call
(``.cpu().numpy()`` at sam_pt.py:830-831, numpy prompt assembly at :726-758, PIL/numpy frame hop at :849).  Here:

  * frames stay on the device; the whole clip is image-encoded in batches (``SamPredictor.encode_frames``) and the
    embeddings stay in HBM;
  * all prompts of the clip are assembled once on the host (the trajectories arrive on the CPU anyway —
    point_tracker/tracker.py:82-83) and uploaded in one copy;
  * each (frame, object) runs ``SamPredictor.track_decode``: the 1-2 prompt passes + R refinement passes +
    IoU-threshold rejection of ``predict_mask`` (sam_pt.py:760-837) as one device-side chain without host syncs.

With a predictor that lacks ``encode_frames``/``track_decode`` (e.g. the CPU oracle in the tests, or upstream's
``SamPredictor``) the same class falls back to the reference's call-by-call protocol (``set_image`` /
``predict_torch``), which is also what the unchanged reference ``SamPt`` does with our predictor.

Not implemented in this round (raise ``NotImplementedError``): point re-initialisation (sam_pt.py:355-543, §8 row f3),
``query_masks`` mode point selection (sam_pt/utils/query_points.py, row f2), patch-similarity filtering (default off).
"""
from __future__ import annotations

from enum import IntEnum
from typing import List, Optional

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


class PointVisibilityType(IntEnum):
    """Visibility codes stored as floats in ``visibilities`` (sam_pt/utils/util.py:267-282)."""
    VISIBLE = 1
    INVISIBLE = 0
    REINIT_FAILED = -1
    OUTSIDE_FRAME = -2
    PATCH_NON_SIMILAR = -3
    REJECTED_AFTER_PATCH_WAS_NON_SIMILAR = -4


class SamPt(nn.Module):
    def __init__(self, point_tracker, sam_predictor, sam_iou_threshold: float,
                 positive_point_selection_method: str = "kmedoids", negative_point_selection_method: str = "mixed",
                 positive_points_per_mask: int = 8, negative_points_per_mask: int = 0,
                 add_other_objects_positive_points_as_negative_points: bool = True,
                 max_other_objects_positive_points: Optional[int] = None, point_tracker_mask_batch_size: int = 5,
                 iterative_refinement_iterations: int = 12, use_patch_matching_filtering: bool = False,
                 patch_size: int = 3, patch_similarity_threshold: float = 0.01, use_point_reinit: bool = False,
                 reinit_point_tracker_horizon: int = 24, reinit_horizon: int = 24,
                 reinit_variant: str = "reinit-at-median-of-area-diff"):
        super().__init__()
        self.point_tracker = point_tracker
        self.sam_predictor = sam_predictor
        self.sam_iou_threshold = sam_iou_threshold
        self._sam = sam_predictor.model  # sam_pt.py:96 (makes .to(device) reach the Sam module when it is one)
        self.iterative_refinement_iterations = iterative_refinement_iterations
        self.positive_point_selection_method = positive_point_selection_method
        self.negative_point_selection_method = negative_point_selection_method
        self.positive_points_per_mask = positive_points_per_mask
        self.negative_points_per_mask = negative_points_per_mask
        self.add_other_objects_positive_points_as_negative_points = add_other_objects_positive_points_as_negative_points
        self.max_other_objects_positive_points = max_other_objects_positive_points
        self.point_tracker_mask_batch_size = point_tracker_mask_batch_size
        self.use_patch_matching_filtering = use_patch_matching_filtering
        self.patch_size, self.patch_similarity_threshold = patch_size, patch_similarity_threshold
        self.use_point_reinit = use_point_reinit
        self.reinit_point_tracker_horizon, self.reinit_horizon = reinit_point_tracker_horizon, reinit_horizon
        self.reinit_variant = reinit_variant
        self.profile = {}

    @property
    def device(self):
        return self._sam.device

    # ------------------------------------------------------------------------------------------------
    def forward(self, video):
        if self.training:
            raise NotImplementedError(f"{self._get_name()} does not support training...")
        images = torch.stack(video["image"], dim=0) if isinstance(video["image"], (list, tuple)) else video["image"]
        n_frames, channels, height, width = images.shape
        assert images.dtype == torch.uint8, "Input images must be in uint8 format (0-255)"
        if video.get("query_masks") is not None:
            raise NotImplementedError("query_masks mode needs query-point selection (sam_pt/utils/query_points.py): "
                                      "SURVEY.md §8 row f2, not built yet")
        if video.get("query_points") is None:
            raise ValueError("No query points or masks provided")
        query_points = video["query_points"]
        n_masks, n_points_per_mask, _ = query_points.shape
        if self.use_point_reinit:
            raise NotImplementedError("point re-initialisation (sam_pt.py:355-543): SURVEY.md §8 row f3, not built yet")
        fused = hasattr(self.sam_predictor, "encode_frames") and hasattr(self.sam_predictor, "track_decode")
        feats = None
        if fused:
            images = images.to(self.device)
            feats = self.sam_predictor.encode_frames(images, chw=True)   # every frame exactly once, embeddings in HBM
        query_masks = self.extract_query_masks(images, query_points, feats)
        assert query_masks.shape == (n_masks, height, width)
        trajectories, visibilities = self._track_points(images, query_points)
        _, logits, scores_per_frame = self._apply_sam_to_trajectories(images, trajectories, visibilities, feats)
        scores = scores_per_frame.mean(dim=0)

        target_hw = tuple(video["target_hw"])
        resize_factor = torch.tensor(target_hw) / torch.tensor(logits.shape[-2:])
        assert (resize_factor[0] - resize_factor[1]).abs().item() < 0.01, "The resizing should have been isotropic"
        if tuple(logits.shape[-2:]) != target_hw:
            logits = self._resize_logits(logits, target_hw)
        trajectories = trajectories * resize_factor
        assert logits.shape == (n_masks, n_frames, target_hw[0], target_hw[1])
        assert trajectories.shape == (n_frames, n_masks, n_points_per_mask, 2)
        assert visibilities.shape == (n_frames, n_masks, n_points_per_mask)
        return {"logits": [m for m in logits], "scores": scores.tolist(), "scores_per_frame": scores_per_frame.tolist(),
                "trajectories": trajectories, "visibilities": visibilities}

    @staticmethod
    def _resize_logits(logits, target_hw):
        """Bilinear (align_corners=False) resize of the (M,T,H,W) logits to target_hw (sam_pt.py:205-206)."""
        if not logits.is_cuda:
            return F.interpolate(logits, size=target_hw, mode="bilinear", align_corners=False)   # reference protocol on CPU
        from . import _lib
        lib = _lib.load()
        M, T, H, W = logits.shape
        out = torch.empty((M, T, target_hw[0], target_hw[1]), dtype=torch.float32, device=logits.device)
        _lib.check(lib.sampt_resize_logits(_lib.ptr(logits.contiguous()), M * T, H, W, _lib.ptr(out), target_hw[0],
                                           target_hw[1], _lib.stream_ptr()), "sampt_resize_logits")
        return out

    # ------------------------------------------------------------------------------------------------
    def extract_query_masks(self, images, query_points, feats=None):
        """SAM on each object's query frame with its own query points (sam_pt.py:308-335).  The reference encodes the
        query frame once per object even when objects share it (App. B-3); with cached embeddings that is a lookup."""
        frame_ids = [int(t.item()) for t in query_points[:, 0, 0]]
        sub_feats = feats[frame_ids] if feats is not None else None
        _, logits, _ = self._apply_sam_to_trajectories(
            images=torch.stack([images[i] for i in frame_ids], dim=0),
            trajectories=query_points[:, None, :, 1:].cpu(),
            visibilities=torch.ones_like(query_points[:, None, :, 0]).cpu(),
            feats=sub_feats)
        return (logits > self.sam_predictor.model.mask_threshold)[0]

    def _track_points(self, rgbs, query_points):
        """Chunks of ``point_tracker_mask_batch_size`` objects per tracker call (sam_pt.py:545-576, 578-692)."""
        if self.use_patch_matching_filtering:
            raise NotImplementedError("patch-similarity filtering (sam_pt.py:597-682) is off by default and not built")
        trajs, viss = [], []
        n_masks = query_points.shape[0]
        rgbs_dev = rgbs.to(self.device).unsqueeze(0)
        self.point_tracker.eval()
        for i in range(0, n_masks, self.point_tracker_mask_batch_size):
            q = query_points[i:i + self.point_tracker_mask_batch_size]
            m, p, _ = q.shape
            with torch.no_grad():
                out = self.point_tracker.to(self.device).evaluate_batch(rgbs_dev, q.reshape(1, m * p, 3).to(self.device))
            t = out["trajectories_pred"].squeeze(0)
            v = out["visibilities_pred"].squeeze(0).float()
            h, w = rgbs.shape[-2:]
            v[t[:, :, 0] / w < 0.01] = PointVisibilityType.OUTSIDE_FRAME.value   # sam_pt.py:684-690
            v[t[:, :, 1] / h < 0.01] = PointVisibilityType.OUTSIDE_FRAME.value
            v[t[:, :, 0] / w > 0.99] = PointVisibilityType.OUTSIDE_FRAME.value
            v[t[:, :, 1] / h > 0.99] = PointVisibilityType.OUTSIDE_FRAME.value
            trajs.append(t.reshape(-1, m, p, 2))
            viss.append(v.reshape(-1, m, p))
        return torch.cat(trajs, dim=1), torch.cat(viss, dim=1)

    # ------------------------------------------------------------------------------------------------
    def _prepare_points(self, trajectories, visibilities, frame_idx, mask_idx, n_masks):
        """Prompt assembly of sam_pt.py:726-758: visible points (== 1), tail points negative, the other objects'
        visible positives appended as negatives."""
        point_coords = trajectories[frame_idx, mask_idx, :, :]
        point_labels = np.ones((len(point_coords)), dtype=int)
        if self.negative_points_per_mask > 0:
            point_labels[self.positive_points_per_mask:] = 0
        vmask = (visibilities[frame_idx, mask_idx, :] == 1).numpy()
        coords = point_coords.numpy()[vmask]
        labels = point_labels[vmask]
        if n_masks > 1 and self.add_other_objects_positive_points_as_negative_points:
            others = [trajectories[frame_idx, o, :self.positive_points_per_mask, :][
                visibilities[frame_idx, o, :self.positive_points_per_mask] == 1, :]
                for o in range(n_masks) if o != mask_idx]
            others = torch.cat(others, dim=0).numpy()
            if self.max_other_objects_positive_points is not None and len(others) > self.max_other_objects_positive_points:
                idx = np.random.choice(len(others), self.max_other_objects_positive_points, replace=False)
                others = others[idx, :]
            coords = np.concatenate([coords, others], axis=0)
            labels = np.concatenate([labels, np.zeros((len(others)), dtype=int)], axis=0)
        return coords, labels

    def _apply_sam_to_trajectories(self, images, trajectories, visibilities, feats=None):
        n_frames, channels, height, width = images.shape
        _, n_masks, points_per_mask, _ = trajectories.shape
        assert trajectories.shape == (n_frames, n_masks, points_per_mask, 2)
        assert visibilities.shape == (n_frames, n_masks, points_per_mask)
        trajectories, visibilities = trajectories.cpu(), visibilities.cpu()
        if feats is not None:
            return self._apply_sam_fused(images, trajectories, visibilities, feats)
        return self._apply_sam_stepwise(images, trajectories, visibilities)

    # -- device-resident path --------------------------------------------------------------------------------
    def _apply_sam_fused(self, images, trajectories, visibilities, feats):
        n_frames, _, height, width = images.shape
        n_masks = trajectories.shape[1]
        dev = self.device
        pred = self.sam_predictor
        size = (height, width)
        prompts = []
        kmax = 1
        for t in range(n_frames):
            for m in range(n_masks):
                c, l = self._prepare_points(trajectories, visibilities, t, m, n_masks)
                if len(c):
                    c = pred.transform.apply_coords(c, size)
                prompts.append((c, l))
                kmax = max(kmax, len(c))
        xy = np.zeros((len(prompts), kmax, 2), dtype=np.float32)
        lab = np.zeros((len(prompts), kmax), dtype=np.int32)
        for i, (c, l) in enumerate(prompts):
            xy[i, :len(c)] = c
            lab[i, :len(c)] = l
        xy_d = torch.from_numpy(xy).to(dev)
        lab_d = torch.from_numpy(lab).to(dev)
        logits = torch.full((n_masks, n_frames, height, width), -float("inf"), dtype=torch.float32, device=dev)
        scores = torch.full((n_frames * n_masks,), -float("inf"), dtype=torch.float32, device=dev)
        two_pass = self.negative_points_per_mask > 0
        # group the (frame, object) items by prompt shape: items of one group run as ONE batched device-side chain
        groups = {}
        for i, (c, l) in enumerate(prompts):
            if len(c) == 0:
                continue                                                         # sam_pt.py:766-767
            key = (len(c), int((l == 1).sum()) if two_pass else -1)
            groups.setdefault(key, []).append(i)
        Fmax = getattr(pred.model, "max_decode_batch", 1)
        for (k, n_pos_first), items in groups.items():
            for s0 in range(0, len(items), Fmax):
                idx = torch.tensor(items[s0:s0 + Fmax], dtype=torch.long, device=dev)
                t_idx, m_idx = idx // n_masks, idx % n_masks
                F_ = idx.numel()
                out_l = torch.empty((F_, height, width), dtype=torch.float32, device=dev)
                out_s = torch.empty((F_,), dtype=torch.float32, device=dev)
                pred.track_decode(feats.index_select(0, t_idx), xy_d.index_select(0, idx).contiguous(),
                                  lab_d.index_select(0, idx).contiguous(), k, n_pos_first,
                                  int(self.iterative_refinement_iterations), float(self.sam_iou_threshold), size,
                                  out_l, out_s)
                logits[m_idx, t_idx] = out_l
                scores[idx] = out_s
        scores_cpu = scores.cpu().view(n_frames, n_masks)                        # the only sync of the SAM stage
        pred_scores = self._mean_scores(scores_cpu)
        return pred_scores, logits, scores_cpu

    @staticmethod
    def _mean_scores(scores_per_frame):
        valid = torch.isfinite(scores_per_frame)
        cnt = valid.sum(0).clamp(min=1)
        return (torch.where(valid, scores_per_frame, torch.zeros_like(scores_per_frame)).sum(0) / cnt).numpy()

    # -- reference protocol (any SamPredictor-compatible object) ---------------------------------------------
    def _apply_sam_stepwise(self, images, trajectories, visibilities):
        n_frames, _, height, width = images.shape
        n_masks = trajectories.shape[1]
        pred = self.sam_predictor
        dev = self.device
        logits = torch.full((n_masks, n_frames, height, width), -float("inf"), dtype=torch.float32)
        scores = torch.full((n_frames, n_masks), -float("inf"), dtype=torch.float32)
        for t in range(n_frames):
            pred.set_image(images[t].permute(1, 2, 0).cpu().numpy())
            for m in range(n_masks):
                c, l = self._prepare_points(trajectories, visibilities, t, m, n_masks)
                if len(c) == 0:
                    continue
                pc = torch.as_tensor(pred.transform.apply_coords(c, pred.original_size), dtype=torch.float, device=dev)
                pl = torch.as_tensor(l, dtype=torch.int, device=dev)
                kw = dict(boxes=None, multimask_output=False, return_logits=True)
                if self.negative_points_per_mask == 0:
                    ml, iou, low = pred.predict_torch(point_coords=pc[None], point_labels=pl[None], mask_input=None, **kw)
                else:
                    _, _, low = pred.predict_torch(point_coords=pc[pl == 1][None], point_labels=pl[pl == 1][None],
                                                   mask_input=None, **kw)
                    ml, iou, low = pred.predict_torch(point_coords=pc[None], point_labels=pl[None], mask_input=low, **kw)
                for _ in range(int(self.iterative_refinement_iterations)):
                    msk = ml[0, 0] > 0
                    if msk.sum() < 2:
                        break
                    yx = msk.nonzero()
                    box = torch.tensor([yx[:, 1].min(), yx[:, 0].min(), yx[:, 1].max(), yx[:, 0].max()],
                                       dtype=torch.float, device=dev)
                    ml, iou, low = pred.predict_torch(point_coords=pc[None], point_labels=pl[None], boxes=box[None, None, :],
                                                      mask_input=low, multimask_output=False, return_logits=True)
                sc = float(iou[0, 0])
                scores[t, m] = sc
                if sc >= self.sam_iou_threshold:
                    logits[m, t] = ml[0, 0].float().cpu()
        return self._mean_scores(scores), logits, scores
