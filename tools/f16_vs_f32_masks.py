#!/usr/bin/env python
"""Mask agreement of the fp16-ViT fast mode against the exact-fp32 mode on the bench workload (same weights, same
inputs, same tracker): reports per-frame IoU statistics.  Random-init weights make this a worst case (logits hover
near zero)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd.point_tracker import PipsPointTracker  # noqa: E402
from sam_pt_amd.sam_predictor import SamHip, SamPredictor  # noqa: E402
from sam_pt_amd.sam_pt import SamPt  # noqa: E402
from sam_pt_amd.synth import bench_clip  # noqa: E402

model_name = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
frames, qp = bench_clip(T=T, seed=72, n_pos=8)
video = {"image": [f.to(dev) for f in frames], "target_hw": tuple(frames.shape[-2:]), "query_points": qp}
res = {}
for prec in ("f32", "f16"):
    sam = SamHip(model_name, precision=prec, seed=72, max_batch=1 if prec == "f32" else 4).to(dev)
    m = SamPt(PipsPointTracker(seed=72), SamPredictor(sam), sam_iou_threshold=-1e9, positive_points_per_mask=8,
              negative_points_per_mask=0, iterative_refinement_iterations=12).eval()
    out = m(video)
    res[prec] = (out["logits"][0] > 0).cpu()
    del m, sam
    torch.cuda.empty_cache()
a, b = res["f32"], res["f16"]
ious = []
for t in range(T):
    u = (a[t] | b[t]).sum().item()
    ious.append(1.0 if u == 0 else (a[t] & b[t]).sum().item() / u)
print(f"{model_name}: f16-vs-f32 mask IoU over {T} frames: min {min(ious):.4f} mean {sum(ious) / T:.4f}; fg fraction f32 {a.float().mean():.3f}")
