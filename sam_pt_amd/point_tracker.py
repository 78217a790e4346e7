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
from typing import Dict, List, Optional, Tuple

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


def _prepared_entry(frames: torch.Tensor, pyr):
    """Cache entry of ``prepare``: holds the frames tensor itself (so its address cannot be recycled for another clip
    while the entry lives) and its version counter (in-place edits invalidate the entry)."""
    return (frames, frames.data_ptr(), tuple(frames.shape), frames._version, pyr)


def _prepared_lookup(entry, frames: torch.Tensor):
    if entry is not None and entry[1] == frames.data_ptr() and entry[2] == tuple(frames.shape) and \
            entry[0]._version == entry[3] and frames._version == entry[3] and entry[0].device == frames.device:
        return entry[4]
    return None


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
        _lib.require_hip(device, "PipsPointTracker")
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
    @_lib.on_device(lambda self, frames, *a, **k: frames.device)
    def compute_pyramid(self, frames: torch.Tensor, chunk_events: Optional[list] = None, shard=None):
        """frames (T,3,H,W) uint8 on device -> list of 4 NHWC f32 levels [T][H_l][W_l][128].  ``chunk_events``: a list that
        receives one ``(first_frame, end_frame, torch.cuda.Event)`` per encoder chunk, recorded on the current stream.
        ``shard`` (sam_pt_amd.dist.FnetShard): encode only this rank's share of the frames and fill in the rest through
        ``shard.exchange`` (all_gather of the pyramid over RCCL); one event for the whole pyramid."""
        self._ensure(frames.device)
        T, _, H, W = frames.shape
        H0, W0 = H // self.stride, W // self.stride
        Tp = T if shard is None else shard.padded_frames(T)
        pyr = [torch.empty((Tp, H0 >> l, W0 >> l, 128), dtype=torch.float32, device=frames.device) for l in range(4)]
        chunk = min(self.fnet_chunk, T)
        nbytes = C.c_size_t()
        _lib.check(self._lib.sampt_pips_fnet_workspace_bytes(self._h, chunk, H, W, C.byref(nbytes)), "fnet_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=frames.device)
        frames = frames.contiguous()

        def encode(lo, hi, events=None):
            for t0 in range(lo, hi, chunk):
                nf = min(chunk, hi - t0)
                outs = _lib.ptr_array([p[t0:t0 + nf] for p in pyr])
                _lib.check(self._lib.sampt_pips_fnet_f32(self._h, _lib.ptr(frames[t0:t0 + nf]), nf, H, W, outs, _lib.ptr(ws),
                                                         nbytes.value, _lib.stream_ptr()), "sampt_pips_fnet_f32")
                if events is not None and frames.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record()
                    events.append((t0, t0 + nf, ev))
                self.stats["fnet_frames"] += nf

        if shard is None:
            encode(0, T, chunk_events)
            return pyr
        mine = shard.mine(T)
        encode(mine.start, mine.stop)
        shard.exchange(pyr, T, encode)
        if chunk_events is not None and frames.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            chunk_events.append((0, T, ev))
        return [p[:T] for p in pyr]

    # -- chained windows for a set of independent point chains (pips/tracker.py:42-153) --------------------------
    def prepare(self, frames: torch.Tensor, shard=None):
        """Optional: build the feature pyramid of ``frames`` (T,3,H,W) now, on the current stream; the next ``forward``
        on the same frames tensor reuses it.  Lets a caller keep the compute-bound fnet on its main stream and run only
        the latency-bound window rounds on a second stream (sam_pt_amd.SamPt).  One event per encoder chunk is kept: a
        ``forward`` running on ANOTHER stream makes every window round wait only for the chunks that hold its frames
        (``chunk_events_on_other_stream``), so the first rounds start while later frames are still being encoded."""
        evs: list = []
        self._prepared = _prepared_entry(frames, self.compute_pyramid(frames, evs, shard=shard))
        self._prepared_events = evs

    chunk_events_on_other_stream = True      # SamPt: the side stream needs no event for the whole pyramid

    def _run_chains(self, pyr, T: int, query_points: torch.Tensor, flipped: torch.Tensor, chunk_events=None):
        """query_points (N,3) CPU float = (t, x, y) in each chain's OWN time axis; ``flipped[i]`` marks chains that run on
        the time-reversed clip (direction frame d = original frame T-1-d).  Returns CPU (T,N,2), (T,N) bool in each
        chain's own time axis.

        The reference walks ``current_frame`` upwards and runs one ``Pips.forward`` per distinct anchor frame with the
        points anchored there (tracker.py:67-109).  PIPS treats points independently (the mixer batch is per point,
        pips.py:525-532), so the same per-point window sequence is executed in ROUNDS: every unfinished chain advances by
        one window per round, all chains batched into one window call.  Results are identical per point; the number of
        window calls drops from one per distinct anchor to max-windows-per-chain.  The whole loop — window frames, write-back
        of frames 1..7, visibility-threshold linking (tracker.py:111-148) — runs on the device inside
        ``sampt_pips_track_f32``: no per-round download / upload, rounds are enqueued one ahead of the GPU."""
        import numpy as np
        dev = pyr[0].device
        N = query_points.shape[0]
        H0, W0 = pyr[0].shape[1:3]
        q_np = np.ascontiguousarray(query_points.numpy().astype(np.float32))
        f_np = np.ascontiguousarray(flipped.numpy().astype(np.uint8))
        q_d, f_d = torch.from_numpy(q_np).to(dev), torch.from_numpy(f_np).to(dev)
        traj = torch.empty((T, N, 2), dtype=torch.float32, device=dev)
        vis = torch.empty((T, N), dtype=torch.float32, device=dev)
        nbytes = C.c_size_t()
        _lib.check(self._lib.sampt_pips_track_workspace_bytes(self._h, N, C.byref(nbytes)), "track_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        evs = list(chunk_events) if chunk_events else []
        ev_arr = (C.c_void_p * max(len(evs), 1))(*[e[2].cuda_event for e in evs])
        lo_arr = (C.c_int * max(len(evs), 1))(*[int(e[0]) for e in evs])
        hi_arr = (C.c_int * max(len(evs), 1))(*[int(e[1]) for e in evs])
        rounds = C.c_int(0)
        _lib.check(self._lib.sampt_pips_track_f32(
            self._h, _lib.ptr_array(pyr), H0, W0, T, N, _lib.ptr(q_d), _lib.ptr(f_d), q_np.ctypes.data_as(C.c_void_p),
            f_np.ctypes.data_as(C.c_void_p), float(self.initial_next_frame_visibility_threshold), 6,
            ev_arr if evs else None, lo_arr if evs else None, hi_arr if evs else None, len(evs), _lib.ptr(traj), _lib.ptr(vis),
            _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), C.byref(rounds)), "sampt_pips_track_f32")
        self.stats["windows"] += rounds.value
        self.stats["launches_per_round"] = int(self._lib.sampt_pips_round_launches(self._h))
        return traj.cpu(), vis.cpu() > 0.5

    @torch.no_grad()
    @_lib.on_device(lambda self, rgbs, query_points: rgbs.device)
    def forward(self, rgbs, query_points):
        from . import prefetch
        prefetch.publish(rgbs[0] if rgbs.dim() == 5 else rgbs)    # unchanged-SamPt fast path: see prefetch.py
        if rgbs.shape[0] != 1:
            raise NotImplementedError("Batch size > 1 is not supported for PIPS yet")  # tracker.py:50-51
        assert rgbs.dtype == torch.uint8, "rgbs must be uint8 (PointTracker.forward contract)"
        dev = rgbs.device
        self._ensure(dev)
        frames = rgbs[0]
        T = frames.shape[0]
        q = query_points[0].detach().float().cpu()
        N = q.shape[0]
        pyr = _prepared_lookup(getattr(self, "_prepared", None), frames)
        chunk_events = getattr(self, "_prepared_events", None) if pyr is not None else None
        if pyr is None:
            pyr = self.compute_pyramid(frames)
        # both temporal directions are independent chains too: run them in the same rounds (tracker.py:159-167)
        qf = q.clone()
        qf[:, 0] = T - qf[:, 0] - 1
        flipped = torch.cat([torch.zeros(N, dtype=torch.bool), torch.ones(N, dtype=torch.bool)])
        # (a direction whose query sits on its last frame never runs a window, tracker.py:67: those chains are left out)
        q_all = torch.cat([q, qf])
        live = (q_all[:, 0] < T - 1).nonzero().flatten()
        tr_all = torch.zeros(T, 2 * N, 2)
        vi_all = torch.zeros(T, 2 * N, dtype=torch.bool)
        tr_all[q_all[:, 0].long(), torch.arange(2 * N)] = q_all[:, 1:]
        vi_all[q_all[:, 0].long(), torch.arange(2 * N)] = True
        if live.numel():
            tr_l, vi_l = self._run_chains(pyr, T, q_all[live], flipped[live], chunk_events)
            tr_all[:, live], vi_all[:, live] = tr_l, vi_l
        for c in (chunk_events or []):                          # chunks no round touched: still order this stream after them
            torch.cuda.current_stream().wait_event(c[2])
        tr_r, vi_r = tr_all[:, :N], vi_all[:, :N]
        tr_l, vi_l = tr_all[:, N:].flip(0), vi_all[:, N:].flip(0)
        traj, vis = tr_r.clone(), vi_r.clone()
        for n in range(N):                                                       # tracker.py:173-199
            s = int(q[n, 0].item())
            traj[:s, n] = tr_l[:s, n]
            vis[:s, n] = vi_l[:s, n]
            assert torch.allclose(traj[s, n], q[n, 1:]) and bool(vis[s, n])
        return traj.unsqueeze(0).to(dev), vis.unsqueeze(0).to(dev)


class PipsPlusPlusPointTracker(PointTracker):
    """PIPS++ behind the reference constructor of
    ``sam_pt.point_tracker.pips_plus_plus.PipsPlusPlusPointTracker`` (tracker.py:11-24; configs/model/point_tracker/
    pips_plus_plus.yaml).  Differences from the reference, on purpose:

    * the encoder runs ONCE per clip; the reference re-runs it for every (query frame, direction, chunk) although its
      per-frame InstanceNorm makes the maps identical, and sub-clips / time-reversed clips become frame-index maps;
    * queries on several frames and on the last frame work (the reference raises IndexError at tracker.py:121 for the
      former and returns T-1 frames for the latter — see oracle/pips2_ref.py);
    * ``image_size`` (off in the shipped config): the bilinear pre-resize runs on the device with torch (plumbing) and
      the float video feeds the encoder directly; the reference's coordinate scaling is kept verbatim, including its
      x <-> height / y <-> width mix-up (tracker.py:77-78, 126-128), which cancels on the way back.
    """

    def __init__(self, checkpoint_path=None, stride=8, max_sequence_length=128, iters=16, image_size=None,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, seed: int = 72, fnet_chunk: int = 8):
        super().__init__()
        from .weights import init_pips2_state_dict
        self.checkpoint_path, self.stride = checkpoint_path, stride
        self.max_sequence_length, self.iters = max_sequence_length, iters
        self.image_size = tuple(image_size) if image_size is not None else None
        sd = state_dict if state_dict is not None else load_pips_checkpoint(checkpoint_path)
        self._sd = sd if sd is not None else init_pips2_state_dict(seed)
        self.fnet_chunk = fnet_chunk
        self._h = None
        self._device = None
        self.stats = {"chunks": 0, "fnet_frames": 0}

    def _ensure(self, device: torch.device):
        if self._h is not None and self._device == device:
            return
        _lib.require_hip(device, "PipsPlusPlusPointTracker")
        from .pack import pack_pips2
        lib = _lib.load()
        self._w = pack_pips2(self._sd, device)
        names, ptrs, n = _lib.name_table(self._w)
        h = C.c_void_p()
        _lib.check(lib.sampt_pips2_create(names, ptrs, n, self.stride, C.byref(h)), "sampt_pips2_create")
        self._h, self._device, self._lib = h, device, lib

    def __del__(self):
        if getattr(self, "_h", None) is not None:
            try:
                self._lib.sampt_pips2_destroy(self._h)
            except Exception:
                pass

    @_lib.on_device(lambda self, frames, *a, **k: frames.device)
    def compute_pyramid(self, frames: torch.Tensor):
        """frames (T,3,H,W) uint8 (or float32 in [0,255]) on device -> 4 NHWC f32 levels [T][H/8 >> l][W/8 >> l][128]."""
        self._ensure(frames.device)
        is_f32 = 1 if frames.dtype == torch.float32 else 0
        T, _, H, W = frames.shape
        H0, W0 = H // self.stride, W // self.stride
        pyr = [torch.empty((T, H0 >> l, W0 >> l, 128), dtype=torch.float32, device=frames.device) for l in range(4)]
        chunk = min(self.fnet_chunk, T)
        nbytes = C.c_size_t()
        _lib.check(self._lib.sampt_pips2_fnet_workspace_bytes(self._h, chunk, H, W, C.byref(nbytes)), "fnet_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=frames.device)
        frames = frames.contiguous()
        for t0 in range(0, T, chunk):
            nf = min(chunk, T - t0)
            outs = _lib.ptr_array([p[t0:t0 + nf] for p in pyr])
            _lib.check(self._lib.sampt_pips2_fnet_f32(self._h, _lib.ptr(frames[t0:t0 + nf]), is_f32, nf, H, W, outs, _lib.ptr(ws),
                                                      ws.numel(), _lib.stream_ptr()), "sampt_pips2_fnet_f32")
        self.stats["fnet_frames"] += T
        return pyr

    def prepare(self, frames: torch.Tensor, shard=None):
        """``shard`` is accepted for interface parity and ignored: PIPS++ encodes the whole clip on every rank."""
        if self.image_size is not None:        # forward() encodes the RESIZED float video: a pyramid of `frames` is unused
            return
        self._prepared = _prepared_entry(frames, self.compute_pyramid(frames))

    def _track(self, pyr, frame_ids: List[int], query_xy: torch.Tensor) -> torch.Tensor:
        """One direction (tracker.py:26-62): the clip is ``frame_ids`` (pyramid frame of every time step); chunks of
        ``max_sequence_length`` frames overlapping by one, templates carried from chunk to chunk.  -> (S,N,2) on device."""
        dev = pyr[0].device
        S_all, N = len(frame_ids), query_xy.shape[0]
        H0, W0 = pyr[0].shape[1:3]
        trajs = query_xy.to(dev)[None].repeat(S_all, 1, 1).contiguous()          # zero-velocity init
        pyr_ptrs = _lib.ptr_array(pyr)
        cur, done, have_init = 0, False, 0
        feats = None
        while not done:
            end = cur + self.max_sequence_length
            if end > S_all:
                diff = end - S_all
                end -= diff
                cur = max(cur - diff, 0)
            S = end - cur
            fidx = torch.tensor(frame_ids[cur:end], dtype=torch.int32)[None].repeat(N, 1).contiguous().to(dev)
            if feats is None:
                feats = [torch.empty((N, S, 128), dtype=torch.float32, device=dev) for _ in range(3)]
            else:                                                                # feat_init[:, :S_local] (tracker.py:52-54)
                feats = [f[:, :S].contiguous() for f in feats]
            nbytes = C.c_size_t()
            _lib.check(self._lib.sampt_pips2_update_workspace_bytes(self._h, N, S, C.byref(nbytes)), "update_workspace")
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            t0 = trajs[cur:end].contiguous()
            out = torch.empty_like(t0)
            _lib.check(self._lib.sampt_pips2_update_f32(self._h, pyr_ptrs, H0, W0, _lib.ptr(fidx), N, S, _lib.ptr(t0),
                                                        have_init, _lib.ptr_array(feats), self.iters, _lib.ptr(out),
                                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                       "sampt_pips2_update_f32")
            trajs[cur:end] = out
            trajs[end:] = trajs[end - 1:end]                                     # zero-velocity for the future
            have_init = 1
            self.stats["chunks"] += 1
            if end >= S_all:
                done = True
            else:
                cur = cur + self.max_sequence_length - 1
        return trajs

    @torch.no_grad()
    @_lib.on_device(lambda self, rgbs, query_points: rgbs.device)
    def forward(self, rgbs, query_points):
        from . import prefetch
        prefetch.publish(rgbs[0] if rgbs.dim() == 5 else rgbs)    # unchanged-SamPt fast path: see prefetch.py
        if rgbs.shape[0] != 1:
            raise NotImplementedError("Only batch size 1 is supported.")          # tracker.py:83
        assert rgbs.dtype == torch.uint8, "rgbs must be uint8 (PointTracker.forward contract)"
        dev = rgbs.device
        self._ensure(dev)
        frames = rgbs[0]
        T, _, H, W = frames.shape
        q = query_points[0].detach().float().cpu()
        prepared = _prepared_lookup(getattr(self, "_prepared", None), frames)
        if self.image_size is not None:                                          # tracker.py:69-78
            fr = torch.nn.functional.interpolate(frames.float() / 255.0, size=self.image_size, mode="bilinear") * 255.0
            pyr = self.compute_pyramid(fr.contiguous())
            q[:, 1] *= self.image_size[0] / H
            q[:, 2] *= self.image_size[1] / W
        elif prepared is not None:
            pyr = prepared
        else:
            pyr = self.compute_pyramid(frames)
        N = q.shape[0]
        groups: Dict[int, List[int]] = {}
        for i in range(N):
            groups.setdefault(int(q[i, 0].item()), []).append(i)
        traj = torch.zeros((T, N, 2), dtype=torch.float32, device=dev)
        for t, idxs in groups.items():                                           # tracker.py:84-122
            xy = q[idxs, 1:].contiguous()
            if t != T - 1:
                traj[t:, idxs] = self._track(pyr, list(range(t, T)), xy)
            if t != 0:
                right = self._track(pyr, list(range(t, -1, -1)), xy).flip(0)     # frames 0..t
                traj[:t + 1 if t == T - 1 else t, idxs] = right if t == T - 1 else right[:-1]
        if self.image_size is not None:                                          # tracker.py:126-128
            traj[:, :, 0] *= H / self.image_size[0]
            traj[:, :, 1] *= W / self.image_size[1]
        vis = torch.ones((1, T, N), dtype=torch.float32, device=dev)             # PIPS++ predicts no visibility (:64)
        return traj.unsqueeze(0), vis


_COTRACKER_MODELS = {"cotracker_stride_4_wind_8": (8, 4), "cotracker_stride_4_wind_12": (12, 4),
                     "cotracker_stride_8_wind_16": (16, 8)}


def cotracker_model_from_checkpoint_name(checkpoint_path: str) -> Tuple[int, int]:
    """(window length S, stride) of a CoTracker checkpoint, chosen from the FILE NAME as upstream's ``build_cotracker`` does
    (co-tracker @ 4f297a9, cotracker/models/build_cotracker.py: ``model_name = checkpoint.split("/")[-1].split(".")[0]``,
    unknown names raise ValueError).  The three checkpoints of configs/model/point_tracker/cotracker.yaml:2-4 share every
    weight SHAPE (UpdateFormer and fnet do not depend on S or the stride), so loading one into the wrong model would run
    silently with the wrong window — the name is the only thing that tells them apart."""
    import os
    name = os.path.basename(str(checkpoint_path)).split(".")[0]
    if name not in _COTRACKER_MODELS:
        raise ValueError(f"Unknown model name {name}")
    return _COTRACKER_MODELS[name]


def load_cotracker_checkpoint(checkpoint_path: Optional[str]):
    """``build_cotracker`` convention (co-tracker @ 4f297a9): a ``.pth`` state dict, optionally under the key 'model'."""
    if checkpoint_path is None:
        return None
    with open(checkpoint_path, "rb") as f:
        sd = torch.load(f, map_location="cpu")
    return sd["model"] if "model" in sd else sd


def get_points_on_a_grid(grid_size: int, interp_shape) -> torch.Tensor:
    """Upstream's support grid (cotracker.py ``get_points_on_a_grid``): (grid_size^2, 2) = (x, y), a regular grid with a
    margin of ``interp_shape[1] // 64`` pixels, rows first."""
    if grid_size == 1:
        return torch.tensor([[interp_shape[1] / 2, interp_shape[0] / 2]])
    step = interp_shape[1] // 64
    lin = torch.linspace(0, grid_size - 1, grid_size)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    gy = step + gy.reshape(-1) / float(grid_size - 1) * (interp_shape[0] - step * 2)
    gx = step + gx.reshape(-1) / float(grid_size - 1) * (interp_shape[1] - step * 2)
    return torch.stack([gx, gy], dim=-1)


class CoTrackerPointTracker(PointTracker):
    """CoTracker behind the constructor of ``sam_pt.point_tracker.cotracker.CoTrackerPointTracker``
    (cotracker/tracker.py:32-66; configs/model/point_tracker/cotracker.yaml) — the reference's default tracker.

    The adapter logic is the reference's: float video resized bilinearly to ``interp_shape`` and queries rescaled
    (tracker.py:87-96), a ``support_grid_size``^2 grid of support queries every ``support_grid_every_n_frames`` frames
    (:98-102), the model with 6 iterations (:104), the time-flipped pass filling entries that are exactly 0 (:154-170),
    support points dropped, visibility > threshold, trajectories scaled back (:144-150); clips shorter than the window
    are padded with their last frame (:12-24).  Different on purpose:

    * the encoder runs ONCE per frame for both temporal directions (upstream re-encodes 4 new frames per window and the
      whole clip again for the flipped pass; InstanceNorm is per-sample, so the maps are identical);
    * a whole direction is ONE device call (``sampt_cotracker_track_f32``): window membership depends only on the query
      frames, the window-to-window carry stays on the device, nothing synchronises with the host;
    * the debug visualisation branch (cv2 / imageio, :107-142) is control plane and not built.
    """

    def __init__(self, checkpoint_path=None, interp_shape=(384, 512), visibility_threshold=0.7, support_grid_size=2,
                 support_grid_every_n_frames=12, add_debug_visualisations=False,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, seed: int = 72, fnet_chunk: int = 8, iters: int = 6):
        super().__init__()
        self.checkpoint_path = checkpoint_path
        self.interp_shape = tuple(interp_shape) if interp_shape is not None else None
        self.visibility_threshold = visibility_threshold
        self.support_grid_size = support_grid_size
        self.support_grid_every_n_frames = support_grid_every_n_frames
        self.add_debug_visualisations = add_debug_visualisations
        if add_debug_visualisations:
            raise NotImplementedError("add_debug_visualisations (cv2 / imageio gif dump, tracker.py:107-142) is not built")
        from .weights import init_cotracker_state_dict
        self.s, self.stride = 8, 4            # the HIP engine is built for cotracker_stride_4_wind_8 (the YAML's default)
        if state_dict is None and checkpoint_path is not None:
            s_ckpt, stride_ckpt = cotracker_model_from_checkpoint_name(checkpoint_path)
            if (s_ckpt, stride_ckpt) != (self.s, self.stride):
                raise NotImplementedError(
                    f"{checkpoint_path}: CoTracker with window {s_ckpt} / stride {stride_ckpt} is not built; the HIP engine "
                    f"implements cotracker_stride_4_wind_8 (window {self.s}, stride {self.stride}) only")
        sd = state_dict if state_dict is not None else load_cotracker_checkpoint(checkpoint_path)
        self._sd = sd if sd is not None else init_cotracker_state_dict(seed)
        self.fnet_chunk, self.iters = fnet_chunk, iters
        self._h = None
        self._device = None
        self._pos: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self.stats = {"windows": 0, "fnet_frames": 0, "calls": 0}

    def _ensure(self, device: torch.device):
        if self._h is not None and self._device == device:
            return
        _lib.require_hip(device, "CoTrackerPointTracker")
        from .pack import pack_cotracker
        lib = _lib.load()
        self._w = pack_cotracker(self._sd, device, self.s)
        names, ptrs, n = _lib.name_table(self._w)
        h = C.c_void_p()
        _lib.check(lib.sampt_cotracker_create(names, ptrs, n, self.stride, self.s, C.byref(h)), "sampt_cotracker_create")
        self._h, self._device, self._lib = h, device, lib
        self._pos = {}

    def __del__(self):
        if getattr(self, "_h", None) is not None:
            try:
                self._lib.sampt_cotracker_destroy(self._h)
            except Exception:
                pass

    # -- resize + encoder for a whole clip ---------------------------------------------------------------------
    @_lib.on_device(lambda self, frames, *a, **k: frames.device)
    def compute_pyramid(self, frames: torch.Tensor, shard=None):
        """frames (T,3,H,W) uint8 / float32 on device -> 4 NHWC f32 levels [T][h/4 >> l][w/4 >> l][128] of the video resized
        to ``interp_shape`` (h, w).  ``shard`` (sam_pt_amd.dist.FnetShard): resize + encode this rank's frames only, the rest
        arrives through ``shard.exchange`` (see PipsPointTracker.compute_pyramid)."""
        self._ensure(frames.device)
        dev = frames.device
        T, _, H, W = frames.shape
        h, w = self.interp_shape if self.interp_shape is not None else (H, W)
        frames = frames.contiguous()
        if frames.dtype != torch.uint8:
            frames = frames.float()
        H0, W0 = h // self.stride, w // self.stride
        Tp = T if shard is None else shard.padded_frames(T)
        pyr = [torch.empty((Tp, H0 >> l, W0 >> l, 128), dtype=torch.float32, device=dev) for l in range(4)]
        chunk = min(self.fnet_chunk, T)
        nbytes = C.c_size_t()
        _lib.check(self._lib.sampt_cotracker_fnet_workspace_bytes(self._h, chunk, h, w, C.byref(nbytes)), "fnet_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)

        def encode(lo, hi):
            if hi <= lo:
                return
            small = torch.empty((hi - lo, 3, h, w), dtype=torch.float32, device=dev)
            _lib.check(self._lib.sampt_resize_frames_f32(_lib.ptr(frames[lo:hi]), 1 if frames.dtype == torch.uint8 else 0,
                                                         (hi - lo) * 3, H, W, _lib.ptr(small), h, w, _lib.stream_ptr()),
                       "sampt_resize_frames_f32")
            for t0 in range(lo, hi, chunk):
                nf = min(chunk, hi - t0)
                outs = _lib.ptr_array([p[t0:t0 + nf] for p in pyr])
                _lib.check(self._lib.sampt_cotracker_fnet_f32(self._h, _lib.ptr(small[t0 - lo:t0 - lo + nf]), nf, h, w, outs,
                                                              _lib.ptr(ws), nbytes.value, _lib.stream_ptr()), "sampt_cotracker_fnet_f32")
            self.stats["fnet_frames"] += hi - lo

        if shard is None:
            encode(0, T)
            return pyr
        mine = shard.mine(T)
        encode(mine.start, mine.stop)
        shard.exchange(pyr, T, encode)
        return [p[:T] for p in pyr]

    def prepare(self, frames: torch.Tensor, shard=None):
        """Build the (resized) clip's feature pyramid now, on the current stream; the next ``forward`` on the same frames
        tensor reuses it (see PipsPointTracker.prepare)."""
        self._prepared = _prepared_entry(frames, self.compute_pyramid(frames, shard=shard))

    def _pos_tables(self, H0: int, W0: int, dev):
        if (H0, W0) not in self._pos:
            from .pack import cotracker_pos_tables
            px, py = cotracker_pos_tables(H0, W0)
            self._pos[(H0, W0)] = (px.to(dev), py.to(dev))
        return self._pos[(H0, W0)]

    def _model(self, pyr, T: int, q: torch.Tensor, flipped: bool):
        """One ``CoTrackerForShortVideosWrapper.__call__`` (tracker.py:17-24): q (n,3) CPU = (t, x, y) in the model's own
        time axis and frame size; ``flipped`` = the clip is time-reversed (model frame t = pyramid frame T-1-t).
        -> traj (T,n,2), vis (T,n) on the device, in the order of q."""
        dev = pyr[0].device
        n = q.shape[0]
        Tm = max(T, self.s)                                            # short clips: last frame repeated
        t = torch.arange(Tm).clamp(max=T - 1)
        fmap = ((T - 1 - t) if flipped else t).to(torch.int32)
        qt = q[:, 0].long()
        order = torch.sort(qt, stable=True).indices                    # CoTracker.forward sorts the points by query frame
        inv = torch.argsort(order)
        qt_s = qt[order].to(torch.int32).contiguous()
        H0, W0 = pyr[0].shape[1:3]
        px, py = self._pos_tables(H0, W0, dev)
        nb = C.c_size_t()
        _lib.check(self._lib.sampt_cotracker_track_workspace_bytes(self._h, n, C.byref(nb)), "track_workspace")
        ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        traj = torch.empty((Tm, n, 2), dtype=torch.float32, device=dev)
        vis = torch.empty((Tm, n), dtype=torch.float32, device=dev)
        fmap_d, qt_d = fmap.to(dev).contiguous(), qt_s.to(dev)
        qxy_d = q[order, 1:].float().contiguous().to(dev)
        _lib.check(self._lib.sampt_cotracker_track_f32(self._h, _lib.ptr_array(pyr), H0, W0, Tm, _lib.ptr(fmap_d), n,
                                                       C.c_void_p(qt_s.data_ptr()), _lib.ptr(qt_d), _lib.ptr(qxy_d),
                                                       _lib.ptr(px), _lib.ptr(py), self.iters, _lib.ptr(traj), _lib.ptr(vis),
                                                       _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                   "sampt_cotracker_track_f32")
        self.stats["calls"] += 1
        inv_d = inv.to(dev)
        return traj[:T].index_select(1, inv_d), vis[:T].index_select(1, inv_d)

    @torch.no_grad()
    @_lib.on_device(lambda self, rgbs, query_points: rgbs.device)
    def forward(self, rgbs, query_points):
        from . import prefetch
        prefetch.publish(rgbs[0] if rgbs.dim() == 5 else rgbs)    # unchanged-SamPt fast path: see prefetch.py
        if rgbs.shape[0] != 1:
            raise NotImplementedError("Batch size > 1 is not supported for CoTracker")   # the model asserts B == 1
        dev = rgbs.device
        self._ensure(dev)
        frames = rgbs[0]
        T, _, H, W = frames.shape
        if self.interp_shape is None:                                   # tracker.py:87-88 (kept on the object)
            self.interp_shape = (H, W)
        h, w = self.interp_shape
        q = query_points[0].detach().float().cpu().clone()
        assert q.shape[1] == 3
        n_points = q.shape[0]
        q[:, 1] *= w / W                                                # tracker.py:95-96
        q[:, 2] *= h / H
        if self.support_grid_size > 0:                                  # tracker.py:98-102
            for i in range(0, T, self.support_grid_every_n_frames):
                g = get_points_on_a_grid(self.support_grid_size, (h, w))
                q = torch.cat([q, torch.cat([torch.full((g.shape[0], 1), float(i)), g], dim=1)], dim=0)
        pyr = _prepared_lookup(getattr(self, "_prepared", None), frames)
        if pyr is None:
            pyr = self.compute_pyramid(frames)
        traj, vis = self._model(pyr, T, q, False)                       # tracker.py:104
        qf = q.clone()                                                  # _compute_backward_tracks, tracker.py:154-170
        qf[:, 0] = T - qf[:, 0] - 1
        traj_f, vis_f = self._model(pyr, T, qf, True)
        traj_f, vis_f = traj_f.flip(0), vis_f.flip(0)
        mask = traj == 0
        traj = torch.where(mask, traj_f, traj)
        vis = torch.where(mask[:, :, 0], vis_f, vis)
        traj, vis = traj[:, :n_points].clone(), vis[:, :n_points].clone()     # tracker.py:144-150
        visb = vis > self.visibility_threshold
        traj[:, :, 0] *= W / float(w)
        traj[:, :, 1] *= H / float(h)
        return traj.unsqueeze(0), visb.unsqueeze(0)
