"""CPU oracle for the PIPS half of the SAM-PT hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 *restatement* of the reference algorithm.  It is imported
only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg; the product
package ``sam_pt_amd`` never imports it (the product fails loudly without its HIP library).

Parity status: **pinned** — ``tests/test_oracle_pins.py`` runs the reference's own ``Pips`` and
``PipsPointTracker`` (imported in place from /root/reference with the namespace-stub recipe of
SURVEY.md Appendix C, see ``oracle/reference_loader.py``) on the same weights/inputs and requires
agreement; ``oracle/make_golden.py`` stores reference outputs under ``tests/golden/``.

Functional style: every function takes the upstream-layout ``state_dict`` (see
``sam_pt_amd/weights.py``) instead of owning ``nn.Module`` parameters.

Reference citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ---------------------------------------------------------------------------------------------
# fnet — BasicEncoder with instance norm (sam_pt/point_tracker/pips/pips.py:191-287, 139-188)
# ---------------------------------------------------------------------------------------------
def _inorm(x):
    # nn.InstanceNorm2d default: affine=False, eps=1e-5, biased variance (pips.py:161-165, 207-209)
    return F.instance_norm(x, eps=1e-5)


def _conv(sd: SD, name: str, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _res_block(sd: SD, p: str, x, stride: int):
    # pips.py:180-188
    y = F.relu(_inorm(_conv(sd, p + ".conv1", x, stride=stride, padding=1)))
    y = F.relu(_inorm(_conv(sd, p + ".conv2", y, padding=1)))
    if stride != 1:
        x = _inorm(_conv(sd, p + ".downsample.0", x, stride=stride))
    return F.relu(x + y)


def fnet(sd: SD, rgbs_norm: torch.Tensor, stride: int = 4, return_scales: bool = False):
    """rgbs_norm: (B,3,H,W) in [-1,1] -> (B,128,H/stride,W/stride).  pips.py:254-287."""
    _, _, H, W = rgbs_norm.shape
    x = F.relu(_inorm(_conv(sd, "fnet.conv1", rgbs_norm, stride=2, padding=3)))
    scales = []
    for li, s in zip((1, 2, 3, 4), (1, 2, 2, 2)):
        x = _res_block(sd, f"fnet.layer{li}.0", x, s)
        x = _res_block(sd, f"fnet.layer{li}.1", x, 1)
        scales.append(x)
    size = (H // stride, W // stride)
    ups = [F.interpolate(t, size, mode="bilinear", align_corners=True) for t in scales]
    y = _conv(sd, "fnet.conv2", torch.cat(ups, dim=1), padding=1)
    y = F.relu(_inorm(y))
    y = _conv(sd, "fnet.conv3", y)
    if return_scales:
        return y, scales
    return y


def normalize_rgbs(rgbs: torch.Tensor) -> torch.Tensor:
    return 2 * (rgbs.float() / 255.0) - 1.0  # pips.py:446


# ---------------------------------------------------------------------------------------------
# correlation pyramid (pips.py:344-407, 320-335)
# ---------------------------------------------------------------------------------------------
def build_pyramid(fmaps: torch.Tensor, levels: int = 4) -> List[torch.Tensor]:
    """fmaps (S,C,H,W) -> list of `levels` maps, each avg_pool2d(2,2) of the previous (pips.py:355-361)."""
    pyr = [fmaps]
    for _ in range(levels - 1):
        pyr.append(F.avg_pool2d(pyr[-1], 2, stride=2))
    return pyr


def corr_volumes(pyr: List[torch.Tensor], ffeats: torch.Tensor) -> List[torch.Tensor]:
    """ffeats (S,N,C) -> per level (S,N,H_l,W_l) = <ffeat, fmap>/sqrt(C)  (pips.py:393-407)."""
    out = []
    for fm in pyr:
        S, C, H, W = fm.shape
        c = torch.matmul(ffeats, fm.reshape(S, C, H * W)).reshape(S, -1, H, W)
        out.append(c / torch.sqrt(torch.tensor(C).float()))
    return out


def sample_corr(vols: List[torch.Tensor], coords: torch.Tensor, radius: int = 3) -> torch.Tensor:
    """coords (S,N,2) in level-0 pixels -> (S,N,levels*(2r+1)^2).  pips.py:364-391.

    Quirk preserved: delta = (dy_i, dx_j) is added to (x, y), i.e. tap (i, j)
    samples at (x + i - r, y + j - r): the x offset varies slowest.
    """
    S, N, _ = coords.shape
    r = radius
    lin = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), dim=-1)  # [i][j] = (lin[i], lin[j])
    outs = []
    for lvl, vol in enumerate(vols):
        H, W = vol.shape[-2:]
        pos = coords.reshape(S * N, 1, 1, 2) / 2 ** lvl + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        gx = 2 * pos[..., 0:1] / (W - 1) - 1  # pips.py:324-326
        gy = 2 * pos[..., 1:2] / (H - 1) - 1
        samp = F.grid_sample(vol.reshape(S * N, 1, H, W), torch.cat([gx, gy], dim=-1), align_corners=True)
        outs.append(samp.view(S, N, -1))
    return torch.cat(outs, dim=-1).contiguous().float()


def bilinear_sample2d(im: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """im (C,H,W), x,y (N,) -> (N,C).  Clamped indices, weights from the unclamped floor
    (sam_pt/point_tracker/utils/samp.py:6-80)."""
    C, H, W = im.shape
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    x1, y1 = x0 + 1, y0 + 1
    cx0, cx1 = x0.clamp(0, W - 1).long(), x1.clamp(0, W - 1).long()
    cy0, cy1 = y0.clamp(0, H - 1).long(), y1.clamp(0, H - 1).long()
    flat = im.permute(1, 2, 0).reshape(H * W, C)
    w00 = ((x1 - x) * (y1 - y))[:, None]
    w01 = ((x - x0) * (y1 - y))[:, None]
    w10 = ((x1 - x) * (y - y0))[:, None]
    w11 = ((x - x0) * (y - y0))[:, None]
    return (w00 * flat[cy0 * W + cx0] + w01 * flat[cy0 * W + cx1]
            + w10 * flat[cy1 * W + cx0] + w11 * flat[cy1 * W + cx1])


# ---------------------------------------------------------------------------------------------
# delta block — sin/cos embedding + MLP-Mixer (pips.py:96-128, 290-317; utils/misc.py:30-55)
# ---------------------------------------------------------------------------------------------
def embed3d(xyz: torch.Tensor, C: int = 64) -> torch.Tensor:
    """(B,S,3) -> (B,S,3C+3): per axis interleaved sin/cos at frequencies arange(0,C,2)*1000/C, then xyz."""
    div = (torch.arange(0, C, 2, dtype=torch.float32) * (1000.0 / C)).reshape(1, 1, C // 2)
    parts = []
    for a in range(3):
        v = xyz[:, :, a:a + 1] * div
        pe = torch.stack([torch.sin(v), torch.cos(v)], dim=-1).reshape(*v.shape[:2], C)
        parts.append(pe)
    return torch.cat(parts + [xyz], dim=2)


def mixer(sd: SD, x: torch.Tensor, S: int = 8, depth: int = 12) -> torch.Tensor:
    """x (B,S,519) -> (B,S*(128+2)).  pips.py:115-128."""
    p = "delta_block.to_delta"
    x = F.linear(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"])
    d = x.shape[-1]
    for i in range(1, depth + 1):
        # token mixing: Conv1d(k=1) with channels = the S tokens (pips.py:116, 120-121)
        y = F.layer_norm(x, (d,), sd[f"{p}.{i}.0.norm.weight"], sd[f"{p}.{i}.0.norm.bias"])
        y = torch.einsum("os,bsc->boc", sd[f"{p}.{i}.0.fn.0.weight"].squeeze(-1), y) + sd[f"{p}.{i}.0.fn.0.bias"][None, :, None]
        y = F.gelu(y)
        y = torch.einsum("so,boc->bsc", sd[f"{p}.{i}.0.fn.3.weight"].squeeze(-1), y) + sd[f"{p}.{i}.0.fn.3.bias"][None, :, None]
        x = y + x
        # channel mixing
        y = F.layer_norm(x, (d,), sd[f"{p}.{i}.1.norm.weight"], sd[f"{p}.{i}.1.norm.bias"])
        y = F.gelu(F.linear(y, sd[f"{p}.{i}.1.fn.0.weight"], sd[f"{p}.{i}.1.fn.0.bias"]))
        y = F.linear(y, sd[f"{p}.{i}.1.fn.3.weight"], sd[f"{p}.{i}.1.fn.3.bias"])
        x = y + x
    x = F.layer_norm(x, (d,), sd[f"{p}.{depth + 1}.weight"], sd[f"{p}.{depth + 1}.bias"])
    x = x.mean(dim=1)
    return F.linear(x, sd[f"{p}.{depth + 3}.weight"], sd[f"{p}.{depth + 3}.bias"])


# ---------------------------------------------------------------------------------------------
# Pips.forward (pips.py:439-620), inference subset
# ---------------------------------------------------------------------------------------------
def pips_forward(sd: SD, xys: torch.Tensor, fmaps: torch.Tensor, feat_init: Optional[torch.Tensor],
                 iters: int = 6, stride: int = 4, S: int = 8, trace: Optional[dict] = None):
    """xys (N,2) px at frame 0 of the window; fmaps (S,128,H/stride,W/stride) of the window's frames.

    Returns (coords_per_iter: list of (S,N,2) px, vis_logits (S,N), ffeat_init (N,128)).
    The dead ``fcp`` accumulation (pips.py:512-519) is skipped.
    """
    N = xys.shape[0]
    C = fmaps.shape[1]
    coords = (xys / float(stride)).reshape(1, N, 2).repeat(S, 1, 1)  # zero-velocity init, pips.py:460-463
    pyr = build_pyramid(fmaps)
    if feat_init is None:
        ffeat = bilinear_sample2d(fmaps[0], coords[0, :, 0], coords[0, :, 1])  # pips.py:469-474
    else:
        ffeat = feat_init
    ffeats = ffeat.unsqueeze(0).repeat(S, 1, 1)  # S,N,C
    coords0 = coords[0].clone()
    preds = []
    for it in range(iters):
        vols = corr_volumes(pyr, ffeats)                     # pips.py:510
        fcorr = sample_corr(vols, coords)                    # S,N,196  pips.py:521
        fcorr_ = fcorr.permute(1, 0, 2)                      # N,S,196
        flows = (coords - coords[0:1]).permute(1, 0, 2)      # N,S,2   pips.py:526
        times = torch.linspace(0, S, S).reshape(1, S, 1).repeat(N, 1, 1)  # pips.py:527 (0, 8/7, ..., 8)
        flows = torch.cat([flows, times], dim=2)
        ffeats_ = ffeats.permute(1, 0, 2)                    # N,S,C
        x = torch.cat([ffeats_, fcorr_, embed3d(flows)], dim=2)            # pips.py:313-314
        delta = mixer(sd, x, S=S).reshape(N, S, C + 2)       # pips.py:315-316
        if trace is not None and it == 0:
            trace["fcorr0"] = fcorr.clone()
            trace["mixer_in0"] = x.clone()
            trace["delta0"] = delta.clone()
        dxy, dfeat = delta[:, :, :2], delta[:, :, 2:]
        g = F.group_norm(dfeat.reshape(N * S, C), 1, sd["norm.weight"], sd["norm.bias"], eps=1e-5)  # pips.py:427,538
        upd = F.gelu(F.linear(g, sd["ffeat_updater.0.weight"], sd["ffeat_updater.0.bias"]))
        ffeats = (upd + ffeats_.reshape(N * S, C)).reshape(N, S, C).permute(1, 0, 2)
        coords = coords + dxy.permute(1, 0, 2)
        coords[0] = coords0                                   # pips.py:543-544
        preds.append(coords * stride)
    vis = F.linear(ffeats.reshape(S * N, C), sd["vis_predictor.0.weight"], sd["vis_predictor.0.bias"]).reshape(S, N)
    return preds, vis, ffeat


# ---------------------------------------------------------------------------------------------
# PipsPointTracker (sam_pt/point_tracker/pips/tracker.py:42-201)
# ---------------------------------------------------------------------------------------------
class PipsTrackerRef:
    """Window chaining / trajectory linking, both directions.  `fnet` is evaluated once per distinct
    frame (legal because InstanceNorm is per-sample: SURVEY.md App. B-7), unless ``cache_fmaps`` is
    False, in which case every window recomputes it exactly like the reference (used by the pin test
    to show both give the same trajectories)."""

    def __init__(self, sd: SD, stride: int = 4, s: int = 8, initial_next_frame_visibility_threshold: float = 0.9,
                 cache_fmaps: bool = True, reference_cost: bool = False):
        """``reference_cost``: spend what the reference spends — ``fnet`` per window (implies ``cache_fmaps=False``)
        and the 6-iteration "init pass" for points starting at a window (tracker.py:81-90) whose only used output is the
        pre-loop feature (App. B-6); results are unchanged, only the time is.  Used by bench.py's cpu_baseline leg."""
        self.sd, self.stride, self.s = sd, stride, s
        self.thr0 = initial_next_frame_visibility_threshold
        self.cache_fmaps = cache_fmaps and not reference_cost
        self.reference_cost = reference_cost
        self.n_windows = 0

    def _fmaps(self, rgbs, idx: List[int], cache: dict):
        """rgbs in ORIGINAL frame order; idx = original frame indices of the window."""
        if not self.cache_fmaps:
            return fnet(self.sd, normalize_rgbs(rgbs[idx]), self.stride)
        miss = [i for i in dict.fromkeys(idx) if i not in cache]
        for i in miss:  # one frame at a time: batch-invariant
            cache[i] = fnet(self.sd, normalize_rgbs(rgbs[i:i + 1]), self.stride)[0]
        return torch.stack([cache[i] for i in idx])

    def _one_direction(self, rgbs: torch.Tensor, query_points: torch.Tensor, cache: dict, index_of):
        """rgbs (T,3,H,W) u8 in ORIGINAL order; `index_of` maps this direction's frame index to the original
        one (identity, or T-1-i for the time-flipped pass); query_points (N,3)=(t,x,y) in direction time."""
        T = rgbs.shape[0]
        N = query_points.shape[0]
        traj = torch.zeros(T, N, 2)
        vis = torch.zeros(T, N)
        start = query_points[:, 0].long()
        ar = torch.arange(N)
        vis[start, ar] = 1.0
        traj[start, ar] = query_points[:, 1:]
        feat_init = torch.zeros(N, 128)
        cur = start.clone()
        for f in range(T - 1):  # tracker.py:67
            active = cur == f
            if active.sum() == 0:
                continue
            idx = list(range(f, min(f + self.s, T)))
            n_missing = self.s - len(idx)
            idx = idx + [idx[-1]] * n_missing                      # tracker.py:73-78
            fm = self._fmaps(rgbs, [index_of(i) for i in idx], cache)
            fresh = start == f
            if fresh.any():                                          # tracker.py:81-90 (App. B-6)
                c = traj[f, fresh] / float(self.stride)
                feat_init[fresh] = bilinear_sample2d(fm[0], c[:, 0], c[:, 1])
                if self.reference_cost:                              # the reference's init pass: a second fnet over the
                    fm_i = self._fmaps(rgbs, [index_of(i) for i in idx], cache)   # window + 6 iterations, discarded
                    pips_forward(self.sd, traj[f, fresh], fm_i, None, iters=6, stride=self.stride, S=self.s)
            preds, vlog, _ = pips_forward(self.sd, traj[f, active], fm, feat_init[active], iters=6,
                                          stride=self.stride, S=self.s)
            self.n_windows += 1
            v = torch.sigmoid(vlog)
            hi = self.s - n_missing
            vis[f + 1:f + hi, active] = v[1:hi]
            traj[f + 1:f + hi, active] = preds[-1][1:hi]
            # linking (tracker.py:111-148)
            thr = torch.where(active, torch.full((N,), self.thr0), torch.zeros(N))
            earliest = torch.where(active, cur + 1, cur)
            last = torch.where(active, cur + hi - 1, cur)
            nxt = last
            while (vis[nxt, ar] <= thr).any():
                nxt = torch.where(vis[nxt, ar] <= thr, nxt - 1, nxt)
                thr = torch.where(nxt < earliest, thr - 0.02, thr)
                nxt = torch.where(nxt < earliest, last, nxt)
            cur = torch.where(active, nxt, cur)
        return traj, vis > 0.5

    @torch.no_grad()
    def forward(self, rgbs: torch.Tensor, query_points: torch.Tensor):
        """rgbs (1,T,3,H,W) u8, query_points (1,N,3) -> (1,T,N,2) f32, (1,T,N) bool.  tracker.py:155-201."""
        assert rgbs.shape[0] == 1
        rgbs = rgbs[0]
        q = query_points[0].float()
        T = rgbs.shape[0]
        cache: dict = {}
        tr_r, vi_r = self._one_direction(rgbs, q, cache, lambda i: i)
        qf = q.clone()
        qf[:, 0] = T - qf[:, 0] - 1
        tr_l, vi_l = self._one_direction(rgbs, qf, cache, lambda i: T - 1 - i)
        tr_l, vi_l = tr_l.flip(0), vi_l.flip(0)
        traj = tr_r.clone()
        vis = vi_r.clone()
        for n in range(q.shape[0]):
            s = int(q[n, 0].item())
            traj[:s, n] = tr_l[:s, n]
            vis[:s, n] = vi_l[:s, n]
        return traj.unsqueeze(0), vis.unsqueeze(0)
