"""Synthetic video + queries for benchmarks and tests (SURVEY.md §8d "synthetic inputs").

A smooth band-limited background translating 2 px/frame and an independently textured foreground disc moving with a
different velocity; queries are positive points at fixed offsets inside the disc (and optional negatives on a ring
outside).  Native 480p clips are upscaled to longest-side 1024 the way demo.py does (nearest, demo.py:210-212), since
the reference pipelines feed SamPt frames whose longest side is already 1024 (configs/demo.yaml:20).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn.functional as F


def synthetic_clip(T: int = 12, H: int = 128, W: int = 256, seed: int = 72, disc_r: float = 20.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns uint8 frames (T,3,H,W) on the CPU and the disc centres (T,2) as (x, y)."""
    g = torch.Generator().manual_seed(seed)
    pad = 2 * T + 8
    bg = F.interpolate(torch.rand(1, 3, (H + pad) // 16 + 2, (W + pad) // 16 + 2, generator=g), size=(H + pad, W + pad),
                       mode="bicubic", align_corners=False)[0].clamp(0, 1)
    side = int(4 * disc_r)
    fg = F.interpolate(torch.rand(1, 3, 8, 8, generator=g), size=(side, side), mode="bicubic", align_corners=False)[0].clamp(0, 1)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    frames, centres = [], []
    for t in range(T):
        f = bg[:, t:t + H, 2 * t:2 * t + W].clone()
        cx, cy = W * 0.35 + 3.0 * t, H * 0.5 + 1.0 * t
        m = ((xx - cx) ** 2 + (yy - cy) ** 2) <= disc_r ** 2
        fy = (yy - cy + 2 * disc_r).clamp(0, side - 1).long()
        fx = (xx - cx + 2 * disc_r).clamp(0, side - 1).long()
        f = torch.where(m[None], fg[:, fy, fx], f)
        frames.append(f)
        centres.append((cx, cy))
    return (torch.stack(frames) * 255).round().to(torch.uint8), torch.tensor(centres)


def disc_queries(centres: torch.Tensor, n_pos: int = 8, r: float = 10.0, t: int = 0) -> torch.Tensor:
    """(n_pos, 3) = (t, x, y) query points inside the disc at frame t."""
    c = centres[t]
    ang = torch.arange(n_pos) * (2 * math.pi / n_pos)
    rad = torch.where(torch.arange(n_pos) % 2 == 0, torch.tensor(float(r)), torch.tensor(float(r) * 0.5))
    xy = torch.stack([c[0] + rad * torch.cos(ang), c[1] + rad * torch.sin(ang)], dim=1)
    return torch.cat([torch.full((n_pos, 1), float(t)), xy], dim=1).float()


def upscale_to_longest_side(frames: torch.Tensor, centres: torch.Tensor, long_side: int = 1024):
    """Nearest-neighbour resize as demo.py:210-212 (F.interpolate on uint8 with the default mode); returns frames and
    centres scaled by the same factor."""
    H, W = frames.shape[-2:]
    s = long_side / max(H, W)
    nh, nw = int(H * s + 0.5), int(W * s + 0.5)
    out = F.interpolate(frames.float(), size=(nh, nw)).to(torch.uint8)
    return out, centres * s


def bench_clip(T: int = 24, seed: int = 72, n_pos: int = 8, n_objects: int = 1, native: bool = False, n_neg: int = 0,
               square: int = 0):
    """The benchmark workload: a 480x854 synthetic clip upscaled to 576x1024 (as the reference pipelines do before SamPt:
    configs/demo.yaml:20), 8 positive query points on one object at t = 0 (BASELINE.json metric: ViT-H + PIPS, 480p,
    8 pts, 1 obj).  ``native`` keeps the 480x854 frames: the tracker then runs at that resolution and SAM resizes."""
    if square:                                               # BASELINE config #5: square input (1024 x 1024), no resize
        frames, centres = synthetic_clip(T=T, H=square, W=square, seed=seed, disc_r=72.0 * square / 1024.0)
        scale = square / 1024.0
    else:
        frames, centres = synthetic_clip(T=T, H=480, W=854, seed=seed, disc_r=60.0)
        scale = 854.0 / 1024.0 if native else 1.0            # query geometry is defined on the 576 x 1024 frames
        if not native:
            frames, centres = upscale_to_longest_side(frames, centres, 1024)
    qs = []
    for m in range(n_objects):           # object 0 = the moving disc; further objects = background patches
        q = disc_queries(centres, n_pos=n_pos, r=36.0 * scale, t=0)
        if n_neg:                        # negatives (tail points, sam_pt.py:731-733) on a ring outside the disc
            q = torch.cat([q, disc_queries(centres, n_pos=n_neg, r=115.0 * scale, t=0)])
        q[:, 1] += 260.0 * scale * (m % 3) * (1 if m < 3 else -1)
        q[:, 2] += (-120.0 if m % 2 else 90.0) * scale * (m > 0)
        qs.append(q)
    return frames, torch.stack(qs)


def bench_query_masks(T: int = 24, seed: int = 72, n_objects: int = 1, native: bool = False, square: int = 0) -> torch.Tensor:
    """Query masks (M, H, W) float {0, 1} at t = 0 for the ``bench_clip`` workload — what a VOS run hands ``SamPt`` instead of
    points (``video["query_masks"]`` + ``query_point_timestep``, sam_pt.py:171-177): object 0 is the moving disc, further
    objects are discs on the background patches ``bench_clip`` puts their query points on."""
    if square:
        H = W = square
        scale, r0 = square / 1024.0, 72.0 * square / 1024.0
        cx, cy = W * 0.35, H * 0.5
    else:
        H, W = (480, 854) if native else (576, 1024)
        scale = 854.0 / 1024.0 if native else 1.0
        up = 1.0 if native else 1024.0 / 854.0
        cx, cy, r0 = 854 * 0.35 * up, 480 * 0.5 * up, 60.0 * up
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    masks = []
    for m in range(n_objects):
        ox = 260.0 * scale * (m % 3) * (1 if m < 3 else -1)
        oy = (-120.0 if m % 2 else 90.0) * scale * (m > 0)
        r = r0 if m == 0 else 50.0 * scale
        masks.append((((xx - cx - ox) ** 2 + (yy - cy - oy) ** 2) <= r * r).float())
    return torch.stack(masks)
