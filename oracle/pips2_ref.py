"""CPU oracle for the PIPS++ point tracker.  TEST INFRASTRUCTURE ONLY.

Imported only by ``tests/`` (and ``oracle/make_golden.py``).  Functional PyTorch-CPU fp32 restatement of
  * ``PipsPlusPlus.forward``      sam_pt/point_tracker/pips_plus_plus/pips_plus_plus.py:436-546
  * ``BasicEncoder`` (stride 8)   :180-260   (identical layer structure to the PIPS encoder -> oracle/pips_ref.fnet)
  * ``DeltaBlock`` / ``ResidualBlock1d`` / ``Conv1dPad``   :12-108, 263-342
  * ``CorrBlock``                 :366-418   (same arithmetic as PIPS' CorrBlock -> oracle/pips_ref)
  * ``posemb_sincos_2d_xy``       sam_pt/point_tracker/utils/misc.py:10-27
  * ``PipsPlusPlusPointTracker``  sam_pt/point_tracker/pips_plus_plus/tracker.py:25-134
driven by a state dict in the reference's key layout (sam_pt_amd/weights.init_pips2_state_dict).

Parity status: **pinned** — ``tests/test_oracle_pins.py`` runs the reference's own ``PipsPlusPlus`` /
``PipsPlusPlusPointTracker`` in place (oracle/reference_loader.load_pips2, build container only) against this file, and
the reference outputs are committed as ``tests/golden/pips2.npz``.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from oracle import pips_ref as P1

SD = Dict[str, torch.Tensor]


def posemb_sincos_2d_xy(xy: torch.Tensor, C: int) -> torch.Tensor:
    """(B,S,2) -> (B,S,C+2): [sin(x w), cos(x w), sin(y w), cos(y w), x, y], w = 10000^-(k/(C/4-1))  (misc.py:10-27)."""
    B, S, _ = xy.shape
    x, y = xy[:, :, 0], xy[:, :, 1]
    omega = torch.arange(C // 4) / (C // 4 - 1)
    omega = 1.0 / (10000 ** omega)
    y = y.flatten()[:, None] * omega[None, :]
    x = x.flatten()[:, None] * omega[None, :]
    pe = torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1).reshape(B, S, C)
    return torch.cat([pe, xy], dim=2)


def _conv1d_same(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Conv1dPad (pips_plus_plus.py:12-40): kernel 3, stride 1 -> one zero on each side."""
    return F.conv1d(F.pad(x, (1, 1)), sd[p + ".conv.weight"], sd[p + ".conv.bias"])


def _res_block(sd: SD, p: str, x: torch.Tensor, cin: int, cout: int, first: bool) -> torch.Tensor:
    """ResidualBlock1d (:43-108) with use_norm=True, use_do=False; InstanceNorm1d has no affine parameters."""
    out = x
    if not first:
        out = F.relu(F.instance_norm(out))
    out = _conv1d_same(sd, p + ".conv1", out)
    out = F.relu(F.instance_norm(out))
    out = _conv1d_same(sd, p + ".conv2", out)
    identity = x
    if cout != cin:
        ch1 = (cout - cin) // 2
        identity = F.pad(identity.transpose(-1, -2), (ch1, cout - cin - ch1)).transpose(-1, -2)
    return out + identity


def delta_block(sd: SD, fcorr: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """DeltaBlock.forward (:318-342): (B,S,588), (B,S,2) -> (B,S,2).  ``first_block_norm`` / ``final_norm`` exist in the
    module but are not applied by the reference's forward."""
    from sam_pt_amd.weights import PIPS2_BLOCKS
    x = torch.cat([fcorr, posemb_sincos_2d_xy(flow, 128)], dim=2).permute(0, 2, 1)
    out = F.relu(_conv1d_same(sd, "delta_block.first_block_conv", x))
    for i, (cin, cout) in enumerate(PIPS2_BLOCKS):
        out = _res_block(sd, f"delta_block.basicblock_list.{i}", out, cin, cout, first=(i == 0))
    out = F.relu(out).permute(0, 2, 1)
    return F.linear(out, sd["delta_block.dense.weight"], sd["delta_block.dense.bias"])


def fnet(sd: SD, rgbs_norm: torch.Tensor, stride: int = 8) -> torch.Tensor:
    """(S,3,H,W) normalised frames -> (S,128,H/8,W/8): the PIPS encoder structure with stride 8 (:180-260)."""
    return P1.fnet(sd, rgbs_norm, stride)


def _sample_frames(fmaps: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """fmaps (S,C,H,W), coords (S,N,2) -> (S,N,C): bilinear_sample2d of frame s at coords[s] (utils/samp.py:6-80)."""
    return torch.stack([P1.bilinear_sample2d(fmaps[s], coords[s, :, 0], coords[s, :, 1]) for s in range(fmaps.shape[0])])


def pips2_forward(sd: SD, trajs_e0: torch.Tensor, fmaps: torch.Tensor, iters: int = 16, stride: int = 8,
                  feat_init: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
    """PipsPlusPlus.forward after the encoder (:456-546).  trajs_e0 (S,N,2) px; fmaps (S,128,H8,W8).
    Returns (list of coordinate predictions (S,N,2) px [iters + 1 entries like ``coord_predictions1``], feats)."""
    S, N, _ = trajs_e0.shape
    coords = trajs_e0.clone() / float(stride)
    if feat_init is not None:
        feats1, feats2, feats4 = feat_init
    else:
        feat1 = P1.bilinear_sample2d(fmaps[0], coords[0, :, 0], coords[0, :, 1])                             # N,C
        feats1 = feat1[None].repeat(S, 1, 1)
        feats2, feats4 = feats1.clone(), feats1.clone()
    coords_bak = coords.clone()
    pyramid = P1.build_pyramid(fmaps)
    preds: List[torch.Tensor] = []
    corr1 = P1.corr_volumes(pyramid, feats1)
    for itr in range(iters):
        if itr >= 1:
            inds2 = (torch.arange(S) - 2).clip(min=0)
            inds4 = (torch.arange(S) - 4).clip(min=0)
            feats2 = _sample_frames(fmaps[inds2], coords[inds2])
            feats4 = _sample_frames(fmaps[inds4], coords[inds4])
        f1 = P1.sample_corr(corr1, coords)                                     # S,N,196
        f2 = P1.sample_corr(P1.corr_volumes(pyramid, feats2), coords)
        f4 = P1.sample_corr(P1.corr_volumes(pyramid, feats4), coords)
        fcorrs = torch.cat([f1, f2, f4], dim=2).permute(1, 0, 2)               # N,S,588
        flows = (coords[1:] - coords[:-1]).permute(1, 0, 2)
        flows = torch.cat([flows, flows[:, -1:]], dim=1)                       # N,S,2
        delta = delta_block(sd, fcorrs, flows)                                 # N,S,2
        coords = coords + delta.permute(1, 0, 2)
        preds.append(coords * stride)                                          # coord_predictions1 (before the lock)
        coords[0] = coords_bak[0]
    preds.append(coords * stride)
    return preds, (feats1, feats2, feats4)


class Pips2TrackerRef:
    """PipsPlusPlusPointTracker (tracker.py:11-134) on the CPU oracle; ``image_size=None`` (configs/model/point_tracker/
    pips_plus_plus.yaml:6).  The encoder is run once per clip (per-frame InstanceNorm -> identical per-frame maps)."""

    def __init__(self, sd: SD, stride: int = 8, max_sequence_length: int = 128, iters: int = 16, image_size=None):
        self.sd, self.stride, self.max_len, self.iters = sd, stride, max_sequence_length, iters
        self.image_size = tuple(image_size) if image_size is not None else None

    def _forward(self, fmaps: torch.Tensor, query_xy: torch.Tensor):
        S = fmaps.shape[0]
        trajs = query_xy[None].repeat(S, 1, 1)
        cur, done, feat_init = 0, False, None
        while not done:
            end = cur + self.max_len
            if end > S:
                diff = end - S
                end -= diff
                cur = max(cur - diff, 0)
            S_local = end - cur
            if feat_init is not None:
                feat_init = tuple(fi[:S_local] for fi in feat_init)
            preds, feat_init = pips2_forward(self.sd, trajs[cur:end], fmaps[cur:end], self.iters, self.stride, feat_init)
            trajs[cur:end] = preds[-1]
            trajs[end:] = trajs[end - 1:end]
            if end >= S:
                done = True
            else:
                cur = cur + self.max_len - 1
        return trajs

    def forward(self, rgbs: torch.Tensor, query_points: torch.Tensor):
        """rgbs (1,T,3,H,W) uint8, query_points (1,N,3)=(t,x,y) -> trajectories (1,T,N,2), visibilities (1,T,N) = 1."""
        T, _, H, W = rgbs.shape[1:]
        frames = rgbs[0]
        query_points = query_points.clone()
        if self.image_size is not None:                                   # tracker.py:69-78 (x <-> H, y <-> W as there)
            frames = F.interpolate(frames.float() / 255.0, size=self.image_size, mode="bilinear") * 255.0
            query_points[:, :, 1] *= self.image_size[0] / H
            query_points[:, :, 2] *= self.image_size[1] / W
        fmaps = fnet(self.sd, P1.normalize_rgbs(frames), self.stride)
        groups = defaultdict(list)
        for idx, pt in enumerate(query_points[0]):
            groups[int(pt[0].item())].append(idx)
        N = query_points.shape[1]
        out = torch.zeros(T, N, 2)
        for t, idxs in groups.items():
            q = query_points[0, idxs, 1:].float()
            left = self._forward(fmaps[t:], q) if t != T - 1 else torch.empty(0, len(idxs), 2)
            right = self._forward(fmaps[:t + 1].flip(0), q).flip(0) if t != 0 else torch.empty(0, len(idxs), 2)
            # (the reference drops the query frame of ``right`` unconditionally, tracker.py:116, and so returns T-1 frames
            #  for a query on the last frame; and it indexes a group's local result with global point indices, :120-122,
            #  which raises IndexError as soon as queries sit on more than one frame.  Both are restated as evidently
            #  intended; the pins only cover what the reference can run: one query frame t < T-1.)
            if t == T - 1:
                out[:, idxs] = right
            else:
                out[:, idxs] = torch.cat([right[:-1], left], dim=0) if len(right) else left
        if self.image_size is not None:                                   # tracker.py:126-128
            out[:, :, 0] *= H / self.image_size[0]
            out[:, :, 1] *= W / self.image_size[1]
        return out[None], torch.ones(1, T, N)
