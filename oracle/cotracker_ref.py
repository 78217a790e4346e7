"""CPU oracle for the CoTracker half of the SAM-PT hot path.  TEST INFRASTRUCTURE ONLY: imported
by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg; never by ``sam_pt_amd``.

Parity status: **parity unpinned for the model, pinned for nothing but structure**.  The adapter
(``CoTrackerTrackerRef.forward``) restates code that IS in the reference tree —
sam_pt/point_tracker/cotracker/tracker.py:12-24 (short-video wrapper), :72-104 (resize to ``interp_shape``, query
rescale, support grid every n frames), :144-150 (drop support points, threshold, rescale) and :154-170 (time-flipped pass,
fill where ``trajectories == 0``).  The MODEL is third-party: facebookresearch/co-tracker @ 4f297a9 (requirements.txt:29),
absent from /root/reference, not installed, and without an independent implementation in this image (transformers has
none).  ``cotracker_forward`` restates its published algorithm (``CoTracker.forward`` / ``forward_iteration``,
``UpdateFormer``, ``AttnBlock`` over timm's ``Attention`` / ``Mlp``, ``get_2d_embedding``,
``get_2d_sincos_pos_embed``, ``sample_pos_embed``, ``get_points_on_a_grid``) from the upstream source as recalled in
SURVEY.md App. A-6, in the checkpoint key layout of ``cotracker_stride_4_wind_8.pth``
(``sam_pt_amd.weights.init_cotracker_state_dict``).  Nothing here has been run against upstream; the shared pieces
(``BasicEncoder``, ``CorrBlock``, ``bilinear_sample2d``) are the PIPS ones, which upstream copied from PIPS and which
ARE pinned (oracle/pips_ref.py).

One point where this restatement deliberately departs from SURVEY.md App. A-6's summary: upstream writes every frame of
a window for every point active in it (``traj_e[:, ind:ind+S, :wind_idx] = coords[-1][:, :S_local]``), so a point's
frames are exact zeros only BEFORE THE FIRST WINDOW that contains its query frame, not before the query frame itself;
the adapter's ``== 0`` back-fill (tracker.py:166-169) therefore covers those frames only.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from oracle import pips_ref as PO

SD = Dict[str, torch.Tensor]
S_WIN = 8
STRIDE = 4
LATENT = 128
HEADS = 8


# ---------------------------------------------------------------------------------------------
# embeddings (cotracker/models/core/embeddings.py)
# ---------------------------------------------------------------------------------------------
def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """get_1d_sincos_pos_embed_from_grid: (M,) -> (M, embed_dim) = [sin(pos * omega), cos(pos * omega)], float64."""
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d_grid(embed_dim: int, H: int, W: int) -> torch.Tensor:
    """get_2d_sincos_pos_embed(embed_dim, (H, W)) -> (H, W, embed_dim) f32: first half encodes the column (``np.meshgrid(
    grid_w, grid_h)`` puts w first), second half the row."""
    gw, gh = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    emb = np.concatenate([sincos_1d(embed_dim // 2, gw), sincos_1d(embed_dim // 2, gh)], axis=1)
    return torch.from_numpy(emb).reshape(H, W, embed_dim).float()


def flow_embedding(xy: torch.Tensor, C: int = 64) -> torch.Tensor:
    """get_2d_embedding(xy, C, cat_coords=True): (..., 2) -> (..., 2 + 2C) = [x, y, pe_x (sin/cos interleaved), pe_y]."""
    x, y = xy[..., 0:1], xy[..., 1:2]
    div = (torch.arange(0, C, 2, dtype=torch.float32) * (1000.0 / C)).reshape(*([1] * (xy.dim() - 1)), C // 2)
    pe_x = torch.zeros(*xy.shape[:-1], C)
    pe_y = torch.zeros(*xy.shape[:-1], C)
    pe_x[..., 0::2], pe_x[..., 1::2] = torch.sin(x * div), torch.cos(x * div)
    pe_y[..., 0::2], pe_y[..., 1::2] = torch.sin(y * div), torch.cos(y * div)
    return torch.cat([xy, pe_x, pe_y], dim=-1)


# ---------------------------------------------------------------------------------------------
# UpdateFormer (cotracker/models/core/cotracker/blocks.py; timm Attention / Mlp)
# ---------------------------------------------------------------------------------------------
def _attn_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """AttnBlock: x + attn(LN(x)); x + mlp(LN(x)).  LayerNorm(elementwise_affine=False, eps=1e-6); timm Attention with
    qkv_bias; Mlp with GELU(approximate='tanh').  x (B, L, C)."""
    B, L, C = x.shape
    hd = C // HEADS
    h = F.layer_norm(x, (C,), eps=1e-6)
    qkv = F.linear(h, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"]).reshape(B, L, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    h = (att @ v).transpose(1, 2).reshape(B, L, C)
    x = x + F.linear(h, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    h = F.layer_norm(x, (C,), eps=1e-6)
    h = F.gelu(F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"]), approximate="tanh")
    return x + F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])


def update_former(sd: SD, x: torch.Tensor, depth: int = 6) -> torch.Tensor:
    """x (N, S, 456) -> (N, S, 130): input_transform, `depth` x (time block over S per point, space block over N per
    frame), flow_head."""
    x = F.linear(x, sd["updateformer.input_transform.weight"], sd["updateformer.input_transform.bias"])
    for i in range(depth):
        x = _attn_block(sd, f"updateformer.time_blocks.{i}", x)                       # batch = points, tokens = frames
        x = _attn_block(sd, f"updateformer.space_blocks.{i}", x.transpose(0, 1)).transpose(0, 1)   # batch = frames
    return F.linear(x, sd["updateformer.flow_head.weight"], sd["updateformer.flow_head.bias"])


# ---------------------------------------------------------------------------------------------
# one window (CoTracker.forward_iteration)
# ---------------------------------------------------------------------------------------------
def forward_iteration(sd: SD, fmaps: torch.Tensor, coords_init: torch.Tensor, feat_init: torch.Tensor,
                      vis_init: torch.Tensor, track_mask: torch.Tensor, iters: int, pos_grid: torch.Tensor,
                      times_embed: torch.Tensor, trace: Optional[dict] = None):
    """fmaps (S,128,H4,W4); coords_init (S,N,2) in feature-map pixels; feat_init (N,128); vis_init (S,N) logits;
    track_mask (S,N) {0,1}.  -> (coords px (S,N,2) after the last iteration, vis logits (S,N))."""
    S, N, _ = coords_init.shape
    coords = coords_init.clone()
    pyr = PO.build_pyramid(fmaps)
    ffeats = feat_init[None].repeat(S, 1, 1)                                          # (S,N,C)
    H4, W4 = fmaps.shape[-2:]
    # sample_pos_embed: the 2-D sin/cos grid embedding sampled (bilinear_sample2d) at the window's first-frame position
    pos = PO.bilinear_sample2d(pos_grid.permute(2, 0, 1), coords[0, :, 0], coords[0, :, 1])   # (N,456)
    for it in range(iters):
        vols = PO.corr_volumes(pyr, ffeats)
        fcorrs = PO.sample_corr(vols, coords)                                         # (S,N,196)
        flows = flow_embedding(coords - coords[0:1])                                  # (S,N,130)
        x = torch.cat([flows, fcorrs, ffeats, track_mask[..., None], vis_init[..., None]], dim=-1)   # (S,N,456)
        x = x + pos[None] + times_embed[:, None]
        if trace is not None and it == 0:
            trace["x0"] = x.clone()
        delta = update_former(sd, x.transpose(0, 1)).transpose(0, 1)                  # (S,N,130)
        if trace is not None and it == 0:
            trace["delta0"] = delta.clone()
        dfeat = delta[..., 2:].reshape(S * N, LATENT)
        dfeat = F.group_norm(dfeat, 1, sd["norm.weight"], sd["norm.bias"])
        upd = F.gelu(F.linear(dfeat, sd["ffeat_updater.0.weight"], sd["ffeat_updater.0.bias"]))
        ffeats = (upd + ffeats.reshape(S * N, LATENT)).reshape(S, N, LATENT)
        coords = coords + delta[..., :2]
    vis = F.linear(ffeats.reshape(S * N, LATENT), sd["vis_predictor.0.weight"], sd["vis_predictor.0.bias"]).reshape(S, N)
    return coords * float(STRIDE), vis


# ---------------------------------------------------------------------------------------------
# CoTracker.forward: sliding windows of S frames with step S/2
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def cotracker_forward(sd: SD, rgbs: torch.Tensor, queries: torch.Tensor, iters: int = 6, fmap_cache: Optional[dict] = None,
                      frame_of=None, trace: Optional[list] = None):
    """rgbs (T,3,H,W) float in [0,255]; queries (N,3) = (t, x, y) px.  -> traj (T,N,2) px (exact zeros where never
    written), vis (T,N) = sigmoid(logit) (0.5 where never written).  ``fmap_cache`` / ``frame_of``: optional per-frame
    fnet cache keyed by ``frame_of(t)`` (InstanceNorm is per-sample, so a frame's map does not depend on its window;
    upstream recomputes the 4 new frames of each window)."""
    T, _, H, W = rgbs.shape
    N = queries.shape[0]
    S = S_WIN
    first = queries[:, 0].long()
    order = torch.sort(first, stable=True).indices            # upstream: torch.sort (ties in unspecified order)
    inv = torch.argsort(order)
    first_s = first[order]
    coords_init = (queries[order, 1:] / float(STRIDE))[None].repeat(S, 1, 1)          # (S,N,2)
    vis_init = torch.full((S, N), 10.0)
    traj = torch.zeros(T, N, 2)
    vis_e = torch.zeros(T, N)
    track_mask = (torch.arange(T)[:, None] >= first_s[None, :]).float()              # (T,N)
    H4, W4 = H // STRIDE, W // STRIDE
    pos_grid = sincos_2d_grid(456, H4, W4)
    times_embed = torch.from_numpy(sincos_1d(456, np.linspace(0, S - 1, S).astype(np.float32))).float()   # (S,456)
    cache = fmap_cache if fmap_cache is not None else {}
    key = frame_of if frame_of is not None else (lambda t: t)

    def fmap(t):
        if key(t) not in cache:
            cache[key(t)] = PO.fnet(sd, PO.normalize_rgbs(rgbs[t:t + 1]), STRIDE)[0]
        return cache[key(t)]

    feat_init = torch.zeros(0, LATENT)
    ind, prev = 0, 0
    coords_prev = vis_prev = None
    while ind < T - S // 2:
        S_local = min(S, T - ind)
        frames = list(range(ind, ind + S_local)) + [ind + S_local - 1] * (S - S_local)   # repeat the last frame
        fm = torch.stack([fmap(t) for t in frames])
        n_act = int((first_s < ind + S).sum())
        if n_act == 0:
            ind += S // 2
            continue
        if n_act > prev:                                       # points that become active in this window
            new = torch.arange(prev, n_act)
            c0 = coords_init[0, new]
            f = torch.stack([PO.bilinear_sample2d(fm[int(first_s[j]) - ind], c0[i:i + 1, 0], c0[i:i + 1, 1])[0]
                             for i, j in enumerate(new.tolist())])
            feat_init = torch.cat([feat_init, f], dim=0)
        if prev > 0:                                           # carry the previous window's second half over
            nc = coords_prev[S // 2:] / float(STRIDE)
            coords_init[:S // 2, :prev] = nc
            coords_init[S // 2:, :prev] = nc[-1:].repeat(S // 2, 1, 1)
            vis_init[:S // 2, :prev] = vis_prev[S // 2:]
            vis_init[S // 2:, :prev] = vis_prev[-1:].repeat(S // 2, 1)
        tm = torch.zeros(S, n_act)
        tm[:S_local] = track_mask[ind:ind + S, :n_act]
        tr = {} if trace is not None else None
        coords, vis = forward_iteration(sd, fm, coords_init[:, :n_act], feat_init[:n_act], vis_init[:, :n_act], tm, iters,
                                        pos_grid, times_embed, tr)
        if trace is not None:
            tr.update(ind=ind, n_act=n_act, coords=coords.clone(), vis=vis.clone())
            trace.append(tr)
        traj[ind:ind + S, :n_act] = coords[:S_local]
        vis_e[ind:ind + S, :n_act] = vis[:S_local]
        track_mask[:ind + S, :n_act] = 0.0
        coords_prev, vis_prev = coords, vis
        ind += S // 2
        prev = n_act
    return traj[:, inv], torch.sigmoid(vis_e[:, inv])


def get_points_on_a_grid(grid_size: int, interp_shape) -> torch.Tensor:
    """(grid_size^2, 2) = (x, y): regular grid with a margin of interp_shape[1] // 64 px, rows first."""
    if grid_size == 1:
        return torch.tensor([[interp_shape[1] / 2, interp_shape[0] / 2]])
    step = interp_shape[1] // 64
    lin = torch.linspace(0, grid_size - 1, grid_size)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    gy = step + gy.reshape(-1) / float(grid_size - 1) * (interp_shape[0] - step * 2)
    gx = step + gx.reshape(-1) / float(grid_size - 1) * (interp_shape[1] - step * 2)
    return torch.stack([gx, gy], dim=-1)


# ---------------------------------------------------------------------------------------------
# CoTrackerPointTracker (sam_pt/point_tracker/cotracker/tracker.py)
# ---------------------------------------------------------------------------------------------
class CoTrackerTrackerRef:
    def __init__(self, sd: SD, interp_shape=(384, 512), visibility_threshold: float = 0.7, support_grid_size: int = 2,
                 support_grid_every_n_frames: int = 12, iters: int = 6):
        self.sd = sd
        self.interp_shape = tuple(interp_shape) if interp_shape is not None else None
        self.visibility_threshold = visibility_threshold
        self.support_grid_size, self.support_grid_every_n_frames = support_grid_size, support_grid_every_n_frames
        self.iters = iters
        self.n_windows = 0

    def _model(self, rgbs, queries, cache, frame_of):
        """CoTrackerForShortVideosWrapper (tracker.py:12-24): clips shorter than S are padded with their last frame."""
        T = rgbs.shape[0]
        if T < S_WIN:
            rgbs = torch.cat([rgbs, rgbs[-1:].repeat(S_WIN - T, 1, 1, 1)], dim=0)
            base = frame_of
            frame_of = lambda t: base(min(t, T - 1))
        tr: list = []
        traj, vis = cotracker_forward(self.sd, rgbs, queries, self.iters, cache, frame_of, trace=tr)
        self.n_windows += len(tr)
        return traj[:T], vis[:T]

    @torch.no_grad()
    def forward(self, rgbs: torch.Tensor, query_points: torch.Tensor):
        """rgbs (1,T,3,H,W) u8/float, query_points (1,N,3) -> (1,T,N,2) f32, (1,T,N) bool.  tracker.py:72-150."""
        assert rgbs.shape[0] == 1
        q = query_points[0].float().clone()
        frames = rgbs[0].float()
        T, _, H, W = frames.shape
        n_points = q.shape[0]
        ishape = self.interp_shape if self.interp_shape is not None else (H, W)
        frames = F.interpolate(frames, tuple(ishape), mode="bilinear")                # tracker.py:91 (align_corners=False)
        q[:, 1] *= ishape[1] / W
        q[:, 2] *= ishape[0] / H
        if self.support_grid_size > 0:                                                # tracker.py:98-102
            for i in range(0, T, self.support_grid_every_n_frames):
                g = get_points_on_a_grid(self.support_grid_size, ishape)
                q = torch.cat([q, torch.cat([torch.full((g.shape[0], 1), float(i)), g], dim=1)], dim=0)
        cache: dict = {}
        traj, vis = self._model(frames, q, cache, lambda t: t)
        # _compute_backward_tracks (tracker.py:154-170)
        qf = q.clone()
        qf[:, 0] = T - qf[:, 0] - 1
        traj_f, vis_f = self._model(frames.flip(0), qf, cache, lambda t: T - 1 - t)
        traj_f, vis_f = traj_f.flip(0), vis_f.flip(0)
        mask = traj == 0
        traj[mask] = traj_f[mask]
        vis[mask[:, :, 0]] = vis_f[mask[:, :, 0]]
        traj, vis = traj[:, :n_points].clone(), vis[:, :n_points].clone()
        visb = vis > self.visibility_threshold
        traj[:, :, 0] *= W / float(ishape[1])
        traj[:, :, 1] *= H / float(ishape[0])
        return traj[None], visb[None]
