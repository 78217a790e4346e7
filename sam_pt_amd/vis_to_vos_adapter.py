"""VIS -> VOS adapter (mirror of sam_pt/modeling/vis_to_vos_adapter.py:17-198).

``SamBasedVisToVosAdapter`` prompts a VOS model (``SamPt``) with SAM's automatic mask proposals of frame 0 and returns
the Mask2Former-style result dict the reference's VIS evaluation consumes.  Same constructor keywords
(configs/vis_eval_root.yaml:8-28), same ``forward(batched_inputs)`` contract and result keys.  The wandb visualisation
branch (vis_to_vos_adapter.py:161-198) is control plane and not built: ``visualize_results`` is accepted and ignored.
"""
from __future__ import annotations

from typing import Any, Dict, List

import torch
from torch import nn


class SamBasedVisToVosAdapter(nn.Module):
    def __init__(self, model, sam_generator, max_num_masks: int, masks_batch_size: int, visualize_results: bool = False,
                 max_videos_to_visualize: int = 0):
        super().__init__()
        self.model = model
        self.sam_generator = sam_generator
        self.max_num_masks = max_num_masks
        self.masks_batch_size = masks_batch_size
        self.visualize_results = False                      # see module docstring
        self.max_videos_to_visualize = max_videos_to_visualize
        self._sam_generator_model = self.sam_generator.predictor.model      # vis_to_vos_adapter.py:58-59

    @property
    def device(self):
        return self._sam_generator_model.device

    @torch.no_grad()
    def forward(self, batched_inputs: List[Dict[str, Any]]) -> Dict[str, Any]:
        """batched_inputs: one dict with ``video_id``, ``image`` (list of uint8 (3,H,W) frames), ``height``, ``width``
        (vis_to_vos_adapter.py:101-121).  Returns image_size, pred_scores, pred_labels, pred_masks, pred_logits,
        trajectories, visibilities (:92-100)."""
        images_list, target_hw, query_masks, query_t, query_labels = self._prepare_query_masks(batched_inputs)
        logits_l, traj_l, vis_l, scores_l = self._track_masks_through_video(query_masks, query_t, images_list, target_hw)
        logits, trajectories, visibilities, scores = self._format_predictions(logits_l, traj_l, vis_l, scores_l)
        return {
            "image_size": target_hw,
            "pred_scores": scores.tolist(),
            "pred_labels": query_labels.tolist(),
            "pred_masks": [m for m in logits > 0],
            "pred_logits": [m for m in logits],
            "trajectories": trajectories,
            "visibilities": visibilities,
        }

    def _prepare_query_masks(self, batched_inputs):
        assert len(batched_inputs) == 1, "Only single video inputs are supported"
        assert batched_inputs[0]["image"][0].dtype == torch.uint8, "Input images must be in uint8 format (0-255)"
        images_list = [i for i in batched_inputs[0]["image"]]
        target_hw = (batched_inputs[0]["height"], batched_inputs[0]["width"])
        records = self.sam_generator.generate(images_list[0].permute(1, 2, 0).cpu().numpy())   # proposals on frame 0
        if not records:
            raise RuntimeError("SAM produced no mask proposals for the first frame")
        dev = self.device
        query_masks = torch.stack([torch.as_tensor(r["segmentation"]) for r in records[:self.max_num_masks]]).to(dev)
        n = query_masks.shape[0]
        query_t = torch.zeros(n, dtype=torch.int64, device=dev)
        query_labels = torch.zeros(n, dtype=torch.int64)        # SAM does not classify its masks
        return images_list, target_hw, query_masks, query_t, query_labels

    def _track_masks_through_video(self, query_masks, query_t, images_list, target_hw):
        logits_l, traj_l, vis_l, scores_l = [], [], [], []
        for i in range(0, query_masks.shape[0], self.masks_batch_size):
            out = self.model({
                "image": images_list,
                "target_hw": target_hw,
                "query_masks": query_masks[i:i + self.masks_batch_size],
                "query_point_timestep": query_t[i:i + self.masks_batch_size],
            })
            logits_l += out["logits"]
            traj_l += out["trajectories"].permute(1, 0, 2, 3)
            vis_l += out["visibilities"].permute(1, 0, 2)
            scores_l += out["scores"]
        assert len(logits_l) == query_masks.shape[0]
        assert tuple(logits_l[0].shape) == (len(images_list),) + tuple(target_hw)
        return logits_l, traj_l, vis_l, scores_l

    @staticmethod
    def _format_predictions(logits_l, traj_l, vis_l, scores_l):
        logits = torch.stack(logits_l, dim=0)                           # (masks, frames, H, W)
        trajectories = torch.stack(traj_l, dim=1)                       # (frames, masks, points, 2)
        visibilities = torch.stack(vis_l, dim=1)
        scores = torch.as_tensor([float(s) for s in scores_l])
        return logits, trajectories, visibilities, scores
