"""Seam 1 of the drop-in boundary: the point-tracker API of the reference, backed by the HIP library.

``PointTracker`` mirrors ``sam_pt.point_tracker.tracker.PointTracker`` (sam_pt/point_tracker/tracker.py:13-118):
same method names, argument meaning and return conventions, so ``SamPt`` (the reference's or ours) can hold either.
``PipsPointTracker`` mirrors ``sam_pt.point_tracker.pips.PipsPointTracker`` (pips/tracker.py:9-201): same constructor
keywords as configs/model/point_tracker/pips.yaml, same window chaining / trajectory linking, but

  * ``fnet`` runs ONCE per distinct frame for the whole clip (InstanceNorm is per-sample, so the per-window
    recompute of the reference is redundant — SURVEY.md App. B-7) and its NHWC feature pyramid stays resident in HBM
    for both temporal directions;
  * every 8-frame window (correlation, MLP-Mixer, feature/coordinate update, visibility head) is one call into
    ``sampt_pips_update_f32``;
  * the redundant "init pass" of the reference (pips/tracker.py:81-90, App. B-6) is a 128-channel bilinear gather.

When the reference package itself is importable, subclass its ABC instead.
"""
from __future__ import annotations

import ctypes as C
import os
from abc import ABC, abstractmethod
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _lib
from .pack import pack_pips
from .weights import init_pips_state_dict


class PointTracker(ABC, nn.Module):
    """Abstract point tracker (interface of sam_pt/point_tracker/tracker.py:13-51)."""

    @abstractmethod
    def forward(self, rgbs, query_points) -> Tuple[torch.Tensor, torch.Tensor]:
        """rgbs (B,T,3,H,W) uint8; query_points (B,N,3)=(t,x,y) -> trajectories (B,T,N,2), visibilities (B,T,N)."""

    def evaluate_batch(self, rgbs, query_points, trajectories_gt=None, visibilities_gt=None):
        # contract of tracker.py:53-89: CPU copies, shape assert
        traj, vis = self.forward(rgbs, query_points)
        assert traj.shape == (rgbs.shape[0], rgbs.shape[1], query_points.shape[1], 2)

        def cpu(t):
            return t.detach().clone().cpu() if t is not None else None

        return {"trajectories_pred": cpu(traj), "visibilities_pred": cpu(vis), "query_points": cpu(query_points),
                "trajectories_gt": cpu(trajectories_gt), "visibilities_gt": cpu(visibilities_gt)}

    @classmethod
    def unpack_results(cls, packed_results, batch_idx):
        out = []
        for b in range(packed_results["trajectories_pred"].shape[0]):
            for n in range(packed_results["trajectories_pred"].shape[2]):
                r = {"idx": f"{batch_idx}_{b}_{n}", "iter": batch_idx, "video_idx": b, "point_idx_in_video": n,
                     "query_point": packed_results["query_points"][b, n, :],
                     "trajectory_pred": packed_results["trajectories_pred"][b, :, n, :],
                     "visibility_pred": packed_results["visibilities_pred"][b, :, n]}
                if packed_results["trajectories_gt"] is not None:
                    r["trajectory_gt"] = packed_results["trajectories_gt"][b, :, n, :]
                    r["visibility_gt"] = packed_results["visibilities_gt"][b, :, n]
                out.append(r)
        return out


def load_pips_checkpoint(checkpoint_path: Optional[str]):
    """`saverloader.load` convention (utils/saverloader.py:30-73): newest model-*.pth in a directory, key
    'model_state_dict'.  None -> seeded random init (no checkpoints exist in this environment)."""
    if checkpoint_path is None:
        return None
    if os.path.isdir(checkpoint_path):
        names = sorted(f for f in os.listdir(checkpoint_path) if f.startswith("model-") and f.endswith(".pth"))
        if not names:
            raise FileNotFoundError(f"no model-*.pth under {checkpoint_path}")
        checkpoint_path = os.path.join(checkpoint_path, names[-1])
    ck = torch.load(checkpoint_path, map_location="cpu")
    return ck.get("model_state_dict", ck)


class PipsPointTracker(PointTracker):
    def __init__(self, checkpoint_path=None, stride=4, s=8, initial_next_frame_visibility_threshold=0.9,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, seed: int = 72, fnet_chunk: int = 8):
        super().__init__()
        self.checkpoint_path, self.stride, self.s = checkpoint_path, stride, s
        self.initial_next_frame_visibility_threshold = initial_next_frame_visibility_threshold
        sd = state_dict if state_dict is not None else load_pips_checkpoint(checkpoint_path)
        self._sd = sd if sd is not None else init_pips_state_dict(seed)
        self.fnet_chunk = fnet_chunk
        self._h = None
        self._w: Dict[str, torch.Tensor] = {}
        self._device = None
        self.stats = {"windows": 0, "fnet_frames": 0}

    # -- engine lifetime ---------------------------------------------------------------------
    def _ensure(self, device: torch.device):
        if self._h is not None and self._device == device:
            return
        if device.type != "cuda":
            raise _lib.SamptError("PipsPointTracker runs on the HIP device only (no CPU fallback); got " + str(device))
        lib = _lib.load()
        self._w = pack_pips(self._sd, device, self.s)
        names, ptrs, n = _lib.name_table(self._w)
        h = C.c_void_p()
        _lib.check(lib.sampt_pips_create(names, ptrs, n, self.stride, self.s, C.byref(h)), "sampt_pips_create")
        self._h, self._device, self._lib = h, device, lib

    def __del__(self):
        if getattr(self, "_h", None) is not None:
            try:
                self._lib.sampt_pips_destroy(self._h)
            except Exception:
                pass

    # -- fnet + pyramid for a whole clip -----------------------------------------------------------
    def compute_pyramid(self, frames: torch.Tensor):
        """frames (T,3,H,W) uint8 on device -> list of 4 NHWC f32 levels [T][H_l][W_l][128]."""
        self._ensure(frames.device)
        T, _, H, W = frames.shape
        H0, W0 = H // self.stride, W // self.stride
        pyr = [torch.empty((T, H0 >> l, W0 >> l, 128), dtype=torch.float32, device=frames.device) for l in range(4)]
        chunk = min(self.fnet_chunk, T)
        nbytes = C.c_size_t()
        _lib.check(self._lib.sampt_pips_fnet_workspace_bytes(self._h, chunk, H, W, C.byref(nbytes)), "fnet_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=frames.device)
        frames = frames.contiguous()
        for t0 in range(0, T, chunk):
            nf = min(chunk, T - t0)
            outs = _lib.ptr_array([p[t0:t0 + nf] for p in pyr])
            _lib.check(self._lib.sampt_pips_fnet_f32(self._h, _lib.ptr(frames[t0:t0 + nf]), nf, H, W, outs, _lib.ptr(ws),
                                                     nbytes.value, _lib.stream_ptr()), "sampt_pips_fnet_f32")
        self.stats["fnet_frames"] += T
        return pyr

    # -- one direction (pips/tracker.py:42-153) ----------------------------------------------------
    def _one_direction(self, pyr, T: int, query_points: torch.Tensor, index_of, ws):
        """query_points (N,3) CPU float (t in this direction's time).  Returns CPU (T,N,2), (T,N) bool."""
        dev = pyr[0].device
        N = query_points.shape[0]
        H0, W0 = pyr[0].shape[1:3]
        traj = torch.zeros(T, N, 2)
        vis = torch.zeros(T, N)
        start = query_points[:, 0].long()
        ar = torch.arange(N)
        vis[start, ar] = 1.0
        traj[start, ar] = query_points[:, 1:]
        feat_init = torch.zeros(N, 128, device=dev)
        cur = start.clone()
        pyr_ptrs = _lib.ptr_array(pyr)
        for f in range(T - 1):
            active = cur == f
            if active.sum() == 0:
                continue
            idx = list(range(f, min(f + self.s, T)))
            n_missing = self.s - len(idx)
            idx = idx + [idx[-1]] * n_missing                                   # tracker.py:73-78
            fidx = torch.tensor([index_of(i) for i in idx], dtype=torch.int32, device=dev)
            fresh = start == f
            if fresh.any():                                                       # tracker.py:81-90 == App. B-6
                xy = (traj[f, fresh] / float(self.stride)).to(dev).contiguous()
                out = torch.empty(int(fresh.sum()), 128, device=dev)
                _lib.check(self._lib.sampt_pips_sample_feat_f32(_lib.ptr(pyr[0][index_of(f)]), H0, W0, _lib.ptr(xy),
                                                                xy.shape[0], _lib.ptr(out), _lib.stream_ptr()),
                           "sampt_pips_sample_feat_f32")
                feat_init[fresh.to(dev)] = out
            n = int(active.sum())
            xys = traj[f, active].to(dev).contiguous()
            fi = feat_init[active.to(dev)].contiguous()
            tr_o = torch.empty(self.s, n, 2, device=dev)
            vi_o = torch.empty(self.s, n, device=dev)
            _lib.check(self._lib.sampt_pips_update_f32(self._h, pyr_ptrs, H0, W0, _lib.ptr(fidx), n, _lib.ptr(xys),
                                                       _lib.ptr(fi), 6, _lib.ptr(tr_o), _lib.ptr(vi_o), _lib.ptr(ws),
                                                       ws.numel(), _lib.stream_ptr()), "sampt_pips_update_f32")
            self.stats["windows"] += 1
            tr_c, vi_c = tr_o.cpu(), vi_o.cpu()       # the linking below is data-dependent host control flow
            hi = self.s - n_missing
            vis[f + 1:f + hi, active] = vi_c[1:hi]
            traj[f + 1:f + hi, active] = tr_c[1:hi]
            # trajectory linking (tracker.py:111-148)
            thr = torch.where(active, torch.full((N,), float(self.initial_next_frame_visibility_threshold)), torch.zeros(N))
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
    def forward(self, rgbs, query_points):
        if rgbs.shape[0] != 1:
            raise NotImplementedError("Batch size > 1 is not supported for PIPS yet")  # tracker.py:50-51
        assert rgbs.dtype == torch.uint8, "rgbs must be uint8 (PointTracker.forward contract)"
        dev = rgbs.device
        self._ensure(dev)
        frames = rgbs[0]
        T = frames.shape[0]
        q = query_points[0].detach().float().cpu()
        N = q.shape[0]
        pyr = self.compute_pyramid(frames)
        nbytes = C.c_size_t()
        _lib.check(self._lib.sampt_pips_update_workspace_bytes(self._h, N, C.byref(nbytes)), "update_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        tr_r, vi_r = self._one_direction(pyr, T, q, lambda i: i, ws)
        qf = q.clone()
        qf[:, 0] = T - qf[:, 0] - 1
        tr_l, vi_l = self._one_direction(pyr, T, qf, lambda i: T - 1 - i, ws)   # time-flipped pass on the same pyramid
        tr_l, vi_l = tr_l.flip(0), vi_l.flip(0)
        traj, vis = tr_r.clone(), vi_r.clone()
        for n in range(N):                                                       # tracker.py:173-199
            s = int(q[n, 0].item())
            traj[:s, n] = tr_l[:s, n]
            vis[:s, n] = vi_l[:s, n]
            assert torch.allclose(traj[s, n], q[n, 1:]) and bool(vis[s, n])
        return traj.unsqueeze(0).to(dev), vis.unsqueeze(0).to(dev)
