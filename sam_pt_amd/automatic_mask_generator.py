"""SAM automatic mask proposals for the VIS adapter.

The reference builds ``segment_anything.automatic_mask_generator.SamAutomaticMaskGenerator`` from
configs/vis_eval_root.yaml:13-28 and calls ``sam_generator.generate(frame0)`` at
sam_pt/modeling/vis_to_vos_adapter.py:113; the first ``max_num_masks`` records' ``"segmentation"`` arrays become the
query masks of the VOS model.  The generator itself is third-party (facebookresearch/segment-anything @ aac76a1,
requirements.txt:26, absent from the reference tree), so this file restates its published algorithm behind the same
constructor keywords and record format:

  point grid (points_per_side^2 points, per crop layer) -> for every batch of ``points_per_batch`` single-point prompts
  ``predict_torch(multimask_output=True, return_logits=True)`` -> keep masks with predicted IoU > ``pred_iou_thresh`` and
  stability score >= ``stability_score_thresh`` (IoU between the logits thresholded at +-``stability_score_offset``) ->
  binarise, bounding boxes, drop masks cut by a crop edge -> box NMS by predicted IoU inside a crop, box NMS across crops
  preferring small crops -> optional small-region clean-up -> records.

MI355X notes: upstream run-length encodes every surviving mask on the host to bound memory; with 288 GB of HBM the boolean
masks simply stay on the device until the records are built (RLE helpers are kept for ``output_mode="uncompressed_rle"``).
All mask arithmetic is device tensor work on the predictor's device; the decoder passes go through the C ABI
(``sampt_sam_decode_multimask``).  Parity: the helpers are pinned against transformers' independent port of the same
utilities (tests/test_oracle_pins.py); box NMS restates torchvision's ``batched_nms`` (absent) and the small-region
clean-up uses ``scipy.ndimage.label`` where upstream uses OpenCV (absent): **those two are parity-unpinned**.
"""
from __future__ import annotations

import math
from itertools import product
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------------------------------------------------
# grid / crop geometry (amg.py: build_point_grid, build_all_layer_point_grids, generate_crop_boxes)
# --------------------------------------------------------------------------------------------------------------------
def build_point_grid(n_per_side: int) -> np.ndarray:
    """(n^2, 2) points (x, y) in [0,1]^2, cell centres, x fastest."""
    offset = 1.0 / (2 * n_per_side)
    side = np.linspace(offset, 1.0 - offset, n_per_side)
    xs, ys = np.meshgrid(side, side)                    # xs[i, j] = side[j], ys[i, j] = side[i]
    return np.stack([xs, ys], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side: int, n_layers: int, scale_per_layer: int) -> List[np.ndarray]:
    """Layer i samples int(n_per_side / scale_per_layer**i) points per side."""
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size: Tuple[int, int], n_layers: int, overlap_ratio: float):
    """XYXY crop boxes: the full image (layer 0), then 2^(i+1) x 2^(i+1) overlapping crops for layer i+1."""
    im_h, im_w = im_size
    short = min(im_h, im_w)
    boxes, layers = [[0, 0, im_w, im_h]], [0]
    for i_layer in range(n_layers):
        n = 2 ** (i_layer + 1)
        overlap = int(overlap_ratio * short * (2 / n))
        crop_w = int(math.ceil((overlap * (n - 1) + im_w) / n))
        crop_h = int(math.ceil((overlap * (n - 1) + im_h) / n))
        x0s = [int((crop_w - overlap) * i) for i in range(n)]
        y0s = [int((crop_h - overlap) * i) for i in range(n)]
        for x0, y0 in product(x0s, y0s):
            boxes.append([x0, y0, min(x0 + crop_w, im_w), min(y0 + crop_h, im_h)])
            layers.append(i_layer + 1)
    return boxes, layers


# --------------------------------------------------------------------------------------------------------------------
# mask utilities (amg.py: calculate_stability_score, batched_mask_to_box, is_box_near_crop_edge, uncrop_*, RLE)
# --------------------------------------------------------------------------------------------------------------------
def calculate_stability_score(logits: torch.Tensor, mask_threshold: float, threshold_offset: float) -> torch.Tensor:
    """|logits > thr + off| / |logits > thr - off| per mask: the IoU of the two nested binarisations."""
    hi = (logits > (mask_threshold + threshold_offset)).flatten(-2).sum(-1, dtype=torch.int32)
    lo = (logits > (mask_threshold - threshold_offset)).flatten(-2).sum(-1, dtype=torch.int32)
    return hi / lo


def batched_mask_to_box(masks: torch.Tensor) -> torch.Tensor:
    """bool (..., H, W) -> int64 (..., 4) XYXY of the set pixels (inclusive max index); [0, 0, 0, 0] for an empty mask."""
    if masks.numel() == 0:
        return torch.zeros(*masks.shape[:-2], 4, dtype=torch.int64, device=masks.device)
    h, w = masks.shape[-2:]
    rows, cols = masks.any(dim=-1), masks.any(dim=-2)                      # (..., H), (..., W)
    ar_h = torch.arange(h, device=masks.device)
    ar_w = torch.arange(w, device=masks.device)
    bottom = (rows * ar_h).amax(dim=-1)
    top = torch.where(rows, ar_h, h).amin(dim=-1)
    right = (cols * ar_w).amax(dim=-1)
    left = torch.where(cols, ar_w, w).amin(dim=-1)
    empty = (right < left) | (bottom < top)
    out = torch.stack([left, top, right, bottom], dim=-1)
    return out * (~empty).unsqueeze(-1)


def is_box_near_crop_edge(boxes: torch.Tensor, crop_box: List[int], orig_box: List[int], atol: float = 20.0):
    """True for crop-frame XYXY boxes with a side within ``atol`` px of a crop edge that is not also an image edge."""
    crop = torch.as_tensor(crop_box, dtype=torch.float, device=boxes.device)
    orig = torch.as_tensor(orig_box, dtype=torch.float, device=boxes.device)
    b = uncrop_boxes_xyxy(boxes, crop_box).float()
    near_crop = (b - crop[None]).abs() <= atol
    near_image = (b - orig[None]).abs() <= atol
    return (near_crop & ~near_image).any(dim=1)


def uncrop_boxes_xyxy(boxes: torch.Tensor, crop_box: List[int]) -> torch.Tensor:
    x0, y0 = crop_box[0], crop_box[1]
    return boxes + torch.tensor([[x0, y0, x0, y0]], device=boxes.device)


def uncrop_points(points: torch.Tensor, crop_box: List[int]) -> torch.Tensor:
    return points + torch.tensor([[crop_box[0], crop_box[1]]], device=points.device)


def uncrop_masks(masks: torch.Tensor, crop_box: List[int], orig_h: int, orig_w: int) -> torch.Tensor:
    x0, y0, x1, y1 = crop_box
    if x0 == 0 and y0 == 0 and x1 == orig_w and y1 == orig_h:
        return masks
    pad_x, pad_y = orig_w - (x1 - x0), orig_h - (y1 - y0)
    return torch.nn.functional.pad(masks, (x0, pad_x - x0, y0, pad_y - y0), value=0)


def box_xyxy_to_xywh(box) -> List[int]:
    b = [int(v) for v in box]
    return [b[0], b[1], b[2] - b[0], b[3] - b[1]]


def mask_to_rle(masks: torch.Tensor) -> List[Dict[str, Any]]:
    """bool (B, H, W) -> uncompressed COCO RLE dicts (column-major runs, the first run counts zeros)."""
    B, h, w = masks.shape
    flat = masks.permute(0, 2, 1).reshape(B, -1).cpu().numpy()
    out = []
    for i in range(B):
        row = flat[i]
        change = np.flatnonzero(row[1:] != row[:-1]) + 1
        edges = np.concatenate([[0], change, [h * w]])
        counts = np.diff(edges).tolist()
        if row[0]:
            counts = [0] + counts
        out.append({"size": [h, w], "counts": counts})
    return out


def rle_to_mask(rle: Dict[str, Any]) -> np.ndarray:
    h, w = rle["size"]
    counts = np.asarray(rle["counts"], dtype=np.int64)
    values = (np.arange(len(counts)) % 2).astype(bool)
    return np.repeat(values, counts).reshape(w, h).T


def area_from_rle(rle: Dict[str, Any]) -> int:
    return int(sum(rle["counts"][1::2]))


# --------------------------------------------------------------------------------------------------------------------
# box NMS (torchvision.ops.batched_nms with a single category; torchvision is absent -> parity unpinned)
# --------------------------------------------------------------------------------------------------------------------
def box_iou_matrix(boxes: torch.Tensor) -> torch.Tensor:
    b = boxes.float()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.maximum(b[:, None, :2], b[None, :, :2])
    rb = torch.minimum(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area[:, None] + area[None, :] - inter)


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS: indices of the kept boxes in order of decreasing score; a box is dropped when its IoU with an already
    kept box is > ``iou_threshold`` (0/0 = nan for degenerate boxes never suppresses, as in torchvision).  The IoU matrix is
    one device op; the greedy sweep over <= a few thousand boxes runs on the host."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    order = torch.argsort(scores.float(), descending=True, stable=True)
    over = (box_iou_matrix(boxes[order]) > iou_threshold).cpu().numpy()
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if not suppressed[i]:
            keep.append(i)
            suppressed |= over[i]
    return order[torch.as_tensor(keep, dtype=torch.int64, device=boxes.device)]


# --------------------------------------------------------------------------------------------------------------------
# small-region clean-up (amg.py: remove_small_regions; OpenCV connected components -> scipy.ndimage.label, 8-connected)
# --------------------------------------------------------------------------------------------------------------------
def remove_small_regions(mask: np.ndarray, area_thresh: float, mode: str) -> Tuple[np.ndarray, bool]:
    """mode "holes": fill background components smaller than ``area_thresh``; "islands": delete such foreground
    components (keeping the largest one if all are small).  Returns (mask, changed)."""
    from scipy import ndimage
    assert mode in ("holes", "islands")
    holes = mode == "holes"
    work = np.logical_xor(holes, mask)
    regions, n = ndimage.label(work, structure=np.ones((3, 3), dtype=bool))
    sizes = np.bincount(regions.ravel(), minlength=n + 1)[1:]
    small = [i + 1 for i, s in enumerate(sizes) if s < area_thresh]
    if not small:
        return mask, False
    fill = [0] + small
    if not holes:
        fill = [i for i in range(n + 1) if i not in fill]
        if not fill:
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


# --------------------------------------------------------------------------------------------------------------------
class _MaskData:
    """Parallel per-mask columns (device tensors) that are filtered and concatenated together."""

    def __init__(self, **cols: torch.Tensor):
        self.cols: Dict[str, torch.Tensor] = dict(cols)

    def __getitem__(self, k):
        return self.cols[k]

    def __setitem__(self, k, v):
        self.cols[k] = v

    def __len__(self):
        return 0 if not self.cols else next(iter(self.cols.values())).shape[0]

    def filter(self, keep: torch.Tensor) -> None:
        for k, v in self.cols.items():
            self.cols[k] = v[keep.to(v.device)]

    def cat(self, other: "_MaskData") -> None:
        for k, v in other.cols.items():
            self.cols[k] = v if k not in self.cols else torch.cat([self.cols[k], v], dim=0)


class SamAutomaticMaskGenerator:
    """Constructor keywords of configs/vis_eval_root.yaml:13-28 (= upstream's).  ``predictor`` (tests) injects any object
    with the ``SamPredictor`` interface instead of building one from ``model``."""

    def __init__(self, model, points_per_side: Optional[int] = 32, points_per_batch: int = 64,
                 pred_iou_thresh: float = 0.88, stability_score_thresh: float = 0.95, stability_score_offset: float = 1.0,
                 box_nms_thresh: float = 0.7, crop_n_layers: int = 0, crop_nms_thresh: float = 0.7,
                 crop_overlap_ratio: float = 512 / 1500, crop_n_points_downscale_factor: int = 1,
                 point_grids: Optional[List[np.ndarray]] = None, min_mask_region_area: int = 0,
                 output_mode: str = "binary_mask", predictor=None) -> None:
        if (points_per_side is None) == (point_grids is None):
            raise ValueError("Exactly one of points_per_side or point_grids must be provided.")
        if points_per_side is not None:
            self.point_grids = build_all_layer_point_grids(points_per_side, crop_n_layers, crop_n_points_downscale_factor)
        else:
            self.point_grids = [np.asarray(g, dtype=np.float64) for g in point_grids]
            if len(self.point_grids) != crop_n_layers + 1:
                raise ValueError("point_grids needs one grid per crop layer (crop_n_layers + 1)")
        if output_mode not in ("binary_mask", "uncompressed_rle", "coco_rle"):
            raise ValueError(f"Unknown output_mode {output_mode}.")
        if output_mode == "coco_rle":
            raise NotImplementedError("output_mode='coco_rle' needs pycocotools, which this build does not ship")
        if isinstance(crop_overlap_ratio, str):                      # the shipped YAML spells it "512 / 1500"
            num, den = crop_overlap_ratio.split("/")
            crop_overlap_ratio = float(num) / float(den)
        if predictor is None:
            from .sam_predictor import SamPredictor
            predictor = SamPredictor(model)
        self.predictor = predictor
        self.points_per_batch = points_per_batch
        self.pred_iou_thresh = pred_iou_thresh
        self.stability_score_thresh = stability_score_thresh
        self.stability_score_offset = stability_score_offset
        self.box_nms_thresh = box_nms_thresh
        self.crop_n_layers = crop_n_layers
        self.crop_nms_thresh = crop_nms_thresh
        self.crop_overlap_ratio = crop_overlap_ratio
        self.crop_n_points_downscale_factor = crop_n_points_downscale_factor
        self.min_mask_region_area = min_mask_region_area
        self.output_mode = output_mode

    @property
    def _device(self):
        return getattr(self.predictor, "device", None) or self.predictor.model.device

    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, image: np.ndarray) -> List[Dict[str, Any]]:
        """image: HxWx3 uint8 RGB.  Returns one record per mask: segmentation (HxW bool array, or an RLE dict), area,
        bbox (XYWH), predicted_iou, point_coords [[x, y]], stability_score, crop_box (XYWH); ordered by decreasing
        predicted IoU within a crop."""
        data = self._generate_masks(image)
        if self.min_mask_region_area > 0 and len(data):
            data = self._postprocess_small_regions(data, self.min_mask_region_area,
                                                   max(self.box_nms_thresh, self.crop_nms_thresh))
        masks = data["masks"].cpu() if len(data) else torch.zeros((0,) + tuple(image.shape[:2]), dtype=torch.bool)
        boxes, crops = data["boxes"].cpu(), data["crop_boxes"].cpu()
        ious, stab, pts = data["iou_preds"].cpu(), data["stability_score"].cpu(), data["points"].cpu()
        rles = mask_to_rle(masks) if self.output_mode != "binary_mask" and len(masks) else None
        records = []
        for i in range(masks.shape[0]):
            records.append({
                "segmentation": masks[i].numpy() if rles is None else rles[i],
                "area": int(masks[i].sum()),
                "bbox": box_xyxy_to_xywh(boxes[i].tolist()),
                "predicted_iou": float(ious[i]),
                "point_coords": [pts[i].tolist()],
                "stability_score": float(stab[i]),
                "crop_box": box_xyxy_to_xywh(crops[i].tolist()),
            })
        return records

    def _empty(self, h: int, w: int) -> _MaskData:
        dev = self._device
        return _MaskData(masks=torch.zeros((0, h, w), dtype=torch.bool, device=dev),
                         iou_preds=torch.zeros(0, device=dev), points=torch.zeros((0, 2), dtype=torch.float64, device=dev),
                         stability_score=torch.zeros(0, device=dev),
                         boxes=torch.zeros((0, 4), dtype=torch.int64, device=dev),
                         crop_boxes=torch.zeros((0, 4), dtype=torch.int64, device=dev))

    def _generate_masks(self, image: np.ndarray) -> _MaskData:
        orig_size = tuple(image.shape[:2])
        crop_boxes, layer_idxs = generate_crop_boxes(orig_size, self.crop_n_layers, self.crop_overlap_ratio)
        data = self._empty(*orig_size)
        for crop_box, layer in zip(crop_boxes, layer_idxs):
            data.cat(self._process_crop(image, crop_box, layer, orig_size))
        if len(crop_boxes) > 1 and len(data):                       # duplicates across crops: the smaller crop wins
            cb = data["crop_boxes"].float()
            scores = 1.0 / ((cb[:, 2] - cb[:, 0]) * (cb[:, 3] - cb[:, 1]))
            data.filter(nms(data["boxes"].float(), scores, self.crop_nms_thresh))
        return data

    def _process_crop(self, image: np.ndarray, crop_box: List[int], layer: int, orig_size: Tuple[int, int]) -> _MaskData:
        x0, y0, x1, y1 = crop_box
        crop = image[y0:y1, x0:x1, :]
        crop_hw = tuple(crop.shape[:2])
        self.predictor.set_image(crop)
        points = self.point_grids[layer] * np.array(crop_hw)[None, ::-1]          # (x, y) in crop pixels
        data = self._empty(*orig_size)
        del data.cols["crop_boxes"]
        for i in range(0, len(points), self.points_per_batch):
            data.cat(self._process_batch(points[i:i + self.points_per_batch], crop_hw, crop_box, orig_size))
        self.predictor.reset_image()
        data.filter(nms(data["boxes"].float(), data["iou_preds"], self.box_nms_thresh))
        data["boxes"] = uncrop_boxes_xyxy(data["boxes"], crop_box)
        data["points"] = uncrop_points(data["points"], crop_box)
        data["crop_boxes"] = torch.tensor([crop_box] * len(data), dtype=torch.int64,
                                          device=data["boxes"].device).reshape(-1, 4)
        return data

    def _process_batch(self, points: np.ndarray, crop_hw: Tuple[int, int], crop_box: List[int],
                       orig_size: Tuple[int, int]) -> _MaskData:
        orig_h, orig_w = orig_size
        dev = self._device
        in_points = torch.as_tensor(self.predictor.transform.apply_coords(points, crop_hw), device=dev)
        in_labels = torch.ones(in_points.shape[0], dtype=torch.int, device=dev)
        logits, iou_preds, _ = self.predictor.predict_torch(in_points[:, None, :].float(), in_labels[:, None],
                                                            multimask_output=True, return_logits=True)
        n_per_point = logits.shape[1]
        data = _MaskData(masks=logits.flatten(0, 1), iou_preds=iou_preds.flatten(0, 1),
                         points=torch.as_tensor(points.repeat(n_per_point, axis=0), device=logits.device))
        del logits
        thr = float(self.predictor.model.mask_threshold)
        if self.pred_iou_thresh > 0.0:
            data.filter(data["iou_preds"] > self.pred_iou_thresh)
        data["stability_score"] = calculate_stability_score(data["masks"], thr, self.stability_score_offset)
        if self.stability_score_thresh > 0.0:
            data.filter(data["stability_score"] >= self.stability_score_thresh)
        data["masks"] = data["masks"] > thr
        data["boxes"] = batched_mask_to_box(data["masks"])
        keep = ~is_box_near_crop_edge(data["boxes"], crop_box, [0, 0, orig_w, orig_h])
        if not bool(keep.all()):
            data.filter(keep)
        data["masks"] = uncrop_masks(data["masks"], crop_box, orig_h, orig_w)
        return data

    @staticmethod
    def _postprocess_small_regions(data: _MaskData, min_area: int, nms_thresh: float) -> _MaskData:
        """Fill holes / drop islands smaller than ``min_area`` pixels, then re-run box NMS preferring untouched masks."""
        dev = data["masks"].device
        masks_np = data["masks"].cpu().numpy()
        new_masks, scores = [], []
        for m in masks_np:
            m, changed_h = remove_small_regions(m, min_area, mode="holes")
            m, changed_i = remove_small_regions(m, min_area, mode="islands")
            new_masks.append(torch.as_tensor(m))
            scores.append(float(not (changed_h or changed_i)))
        masks = torch.stack(new_masks).to(dev)
        boxes = batched_mask_to_box(masks)
        scores_t = torch.as_tensor(scores, device=dev)
        keep = nms(boxes.float(), scores_t, nms_thresh)
        changed = scores_t == 0.0
        data["masks"] = torch.where(changed[:, None, None], masks, data["masks"])
        data["boxes"] = torch.where(changed[:, None], boxes, data["boxes"])
        data.filter(keep)
        return data
