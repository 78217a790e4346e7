"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes as C

import numpy as np
import torch


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def max_abs(a, b) -> float:
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def iou(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.cpu().bool(), b.cpu().bool()
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u


def synthetic_clip(T=12, H=128, W=256, seed=72, disc_r=20):
    """Smooth band-limited background translating 2 px/frame + an independently textured moving disc
    (SURVEY.md §8d).  Returns uint8 (T,3,H,W) and the disc centres (T,2)."""
    g = torch.Generator().manual_seed(seed)
    pad = 2 * T + 8
    bg = torch.nn.functional.interpolate(torch.rand(1, 3, (H + pad) // 16 + 2, (W + pad) // 16 + 2, generator=g),
                                         size=(H + pad, W + pad), mode="bicubic", align_corners=False)[0].clamp(0, 1)
    fg = torch.nn.functional.interpolate(torch.rand(1, 3, 8, 8, generator=g), size=(4 * disc_r, 4 * disc_r),
                                         mode="bicubic", align_corners=False)[0].clamp(0, 1)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    frames, centres = [], []
    for t in range(T):
        f = bg[:, t:t + H, 2 * t:2 * t + W].clone()
        cx, cy = W * 0.35 + 3.0 * t, H * 0.5 + 1.0 * t
        m = ((xx - cx) ** 2 + (yy - cy) ** 2) <= disc_r ** 2
        fy = (yy - cy + 2 * disc_r).clamp(0, 4 * disc_r - 1).long()
        fx = (xx - cx + 2 * disc_r).clamp(0, 4 * disc_r - 1).long()
        tex = fg[:, fy, fx]
        f = torch.where(m[None], tex, f)
        frames.append(f)
        centres.append((cx, cy))
    return (torch.stack(frames) * 255).round().to(torch.uint8), torch.tensor(centres)


def disc_queries(centres, n_pos=8, r=10.0, t=0):
    c = centres[t]
    ang = torch.arange(n_pos) * (2 * np.pi / n_pos)
    rad = torch.where(torch.arange(n_pos) % 2 == 0, torch.tensor(r), torch.tensor(r * 0.5))
    xy = torch.stack([c[0] + rad * torch.cos(ang), c[1] + rad * torch.sin(ang)], dim=1)
    return torch.cat([torch.full((n_pos, 1), float(t)), xy], dim=1).float()
