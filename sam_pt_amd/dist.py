"""Multi-GPU sharding of the SAM-PT hot path: one process per GPU.

* Sequences are independent: in the default mode (one clip per rank, ``lpt_assign`` for sets of unequal clips) ranks never
  exchange activations; the only communication is the final gather of the uint8 index masks to rank 0 (0.59 MB per 576x1024
  frame), which replaces the reference's pickled-RLE ``comm.gather``
  (vis_eval/mask2former_video/data_video/ytvis_eval.py:119-131).
* ONE clip over all ranks (``sharded_forward``, BASELINE config #5): the SAM stage (image encoder + mask decoder) is per frame
  and is dealt out in frame batches; the tracker's per-frame encoder is frame-sharded too and the ranks ``all_gather`` its
  feature pyramid (``FnetShard`` — the one real exchange step); the tracker's window chain is sequential in time by
  construction (pips/tracker.py:67-148: a window starts where the previous one ended) and is replicated.

On ROCm the ``nccl`` backend is RCCL over xGMI; every collective here is one fixed-shape call.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():            # one process per GPU, whatever the backend (gloo on a GPU box included)
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def host_threads(world: int, cpus: Optional[int] = None) -> int:
    """PyTorch host threads for ONE of `world` rank processes sharing a node: weight generation / packing and the prompt assembly
    are host work, and N ranks each spawning a thread per core collapse (PyTorch-CPU oversubscription on the 256-thread GPU hosts).
    At most 32, at least 1, never more than an equal share of the cores."""
    import os
    cpus = cpus or os.cpu_count() or 8
    return max(1, min(32, cpus // max(int(world), 1)))


def lpt_assign(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of sequences to ranks (balanced version of detectron2's contiguous
    InferenceSampler, vis_eval/mask2former_video/data_video/build.py:222-229).  Returns per-rank lists of indices."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += lengths[i]
    return out


# DAVIS-2017 val: 30 sequences, 34-104 frames [recall; the dataset is absent here] — used by bench.py --shard lpt to put
# `lpt_assign` (the answer to the length imbalance of BASELINE config #4) on a timed path with synthetic clips.
DAVIS17_VAL_LENGTHS = [69, 50, 80, 84, 90, 75, 40, 104, 90, 60, 66, 52, 50, 90, 78, 50, 81, 34, 50, 47, 49, 50, 79, 40, 80,
                       100, 79, 43, 40, 99]


def frame_shares(n_frames: int, world: int) -> List[range]:
    """Contiguous equal shares of a clip's frames (the last may be short or empty): rank r encodes ``frame_shares(T, N)[r]``
    with the tracker's encoder."""
    per = -(-n_frames // world)
    return [range(min(r * per, n_frames), min((r + 1) * per, n_frames)) for r in range(world)]


class FnetShard:
    """Frame-sharded tracker encoder for ONE clip over all ranks (BASELINE config #5).

    ``fnet`` (BasicEncoder, pips.py:139-188) is per frame — InstanceNorm normalises each sample on its own (App. B-7) — so rank r
    encodes only its share of the frames and the ranks ``all_gather`` the 4-level feature pyramid (25 MB per 576x1024 frame in
    fp32: [T][H/4 >> l][W/4 >> l][128], l = 0..3).  This is the one real exchange step of the in-clip mode: every rank then
    holds the pyramid a single GPU would have computed, bit for bit, and runs the (latency-bound, sequential) window chain on it.
    On ROCm the ``nccl`` backend is RCCL over xGMI; one ``all_gather_into_tensor`` per level, in place on the level's buffer.

    ``emulate=True`` (bench.py --emulate-rank r/N, one GPU, no process group): the exchange is replaced by computing the other
    ranks' frames locally between two events (``stub_ms`` = their GPU time, to be subtracted from the step) while
    ``bytes_received`` records what the collective would have delivered."""

    def __init__(self, rank: int, world: int, emulate: bool = False, group=None):
        self.rank, self.world, self.emulate, self.group = rank, world, emulate, group
        self.stub_events: list = []
        self.bytes_received = 0

    def padded_frames(self, T: int) -> int:
        return -(-T // self.world) * self.world

    def mine(self, T: int) -> range:
        return frame_shares(T, self.world)[self.rank]

    def exchange(self, pyr: List[torch.Tensor], T: int, compute_range=None) -> None:
        """pyr: the levels, each [padded_frames(T)][h][w][C] with this rank's share already computed (on the current stream)."""
        per = self.padded_frames(T) // self.world
        lo = self.rank * per
        self.bytes_received += sum(int(p[0].numel()) * p.element_size() * (T - len(self.mine(T))) for p in pyr)
        if self.emulate:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for r, share in enumerate(frame_shares(T, self.world)):
                if r != self.rank and len(share):
                    compute_range(share.start, share.stop)
            b.record()
            self.stub_events.append((a, b))
            return
        for p in pyr:
            dist.all_gather_into_tensor(p, p[lo:lo + per].clone(), group=self.group)

    def stub_ms(self) -> float:
        """GPU milliseconds of the emulation's stand-in work since the last call (synchronises)."""
        ms = 0.0
        for a, b in self.stub_events:
            b.synchronize()
            ms += a.elapsed_time(b)
        self.stub_events = []
        return ms


def frame_batches(n_frames: int, world: int, rank: int, batch: int = 8) -> List[range]:
    """Config #5 style sharding inside one sequence: frame batches b = rank (mod world)."""
    starts = list(range(0, n_frames, batch))
    return [range(s, min(s + batch, n_frames)) for i, s in enumerate(starts) if i % world == rank]


def index_masks(logits: torch.Tensor, query_timestep=None, query_masks=None, out_hw=None) -> torch.Tensor:
    """(M,T,H,W) per-object logits -> uint8 (T,H,W) object index map with background 0 (bg logit 0 stacked in front,
    softmax/argmax over objects: sam_pt/vos_eval/eval.py:304, 326, 355).  With ``query_timestep`` (M,) the evaluator's
    overrides are applied first (eval.py:318-323): object m is -1e8 before its query frame, and on the query frame its
    logits are +-1e8 from ``query_masks`` (M,H,W) {0,1} when given (resized with ``nearest`` to (H,W) by the caller).
    ``out_hw`` = the evaluator's resize back to the original frame size (eval.py:340-356): the softmax PROBABILITIES are
    interpolated bilinearly (align_corners=False) before the argmax."""
    from . import _lib
    with _lib.device_guard(logits.device):
        return _index_masks(logits, query_timestep, query_masks, out_hw)


def _index_masks(logits: torch.Tensor, query_timestep=None, query_masks=None, out_hw=None) -> torch.Tensor:
    """Implementation of ``index_masks`` (runs with the HIP device of ``logits`` current)."""
    M, T, H, W = logits.shape
    if out_hw is not None and tuple(out_hw) != (H, W):
        out_hw = (int(out_hw[0]), int(out_hw[1]))
        qt = torch.zeros(M, dtype=torch.int32) if query_timestep is None else torch.as_tensor(query_timestep).to(torch.int32)
        if logits.is_cuda:
            from . import _lib
            lib = _lib.load()
            logits = logits.contiguous()
            out = torch.empty((T,) + out_hw, dtype=torch.uint8, device=logits.device)
            qt_d = qt.to(logits.device).contiguous()
            gt = None if query_masks is None else (query_masks.to(logits.device) > 0).to(torch.uint8).contiguous()
            _lib.check(lib.sampt_vos_index_masks_resized(_lib.ptr(logits), M, T, H, W, _lib.ptr(qt_d), _lib.ptr(gt),
                                                         out_hw[0], out_hw[1], _lib.ptr(out), _lib.stream_ptr()),
                       "sampt_vos_index_masks_resized")
            return out
        lg = logits.clone()                                                       # host-side formula (CPU tests)
        for m in range(M):
            t = int(qt[m])
            lg[m, :t] = -1e8
            if query_masks is not None:
                lg[m, t] = torch.where(query_masks[m] > 0, 1e8, -1e8)
        prob = torch.softmax(torch.cat([torch.zeros((1, T, H, W), dtype=lg.dtype), lg], dim=0), dim=0)   # (M+1,T,H,W)
        prob = torch.nn.functional.interpolate(prob.permute(1, 0, 2, 3), out_hw, mode="bilinear", align_corners=False)
        return prob.argmax(dim=1).to(torch.uint8)
    if logits.is_cuda:
        from . import _lib
        lib = _lib.load()
        logits = logits.contiguous()
        out = torch.empty((T, H, W), dtype=torch.uint8, device=logits.device)
        if query_timestep is None:
            _lib.check(lib.sampt_index_masks(_lib.ptr(logits), M, T * H * W, _lib.ptr(out), _lib.stream_ptr()),
                       "sampt_index_masks")
        else:
            qt = torch.as_tensor(query_timestep).to(logits.device, torch.int32).contiguous()
            gt = None if query_masks is None else (query_masks.to(logits.device) > 0).to(torch.uint8).contiguous()
            _lib.check(lib.sampt_vos_index_masks(_lib.ptr(logits), M, T, H * W, _lib.ptr(qt), _lib.ptr(gt), _lib.ptr(out),
                                                 _lib.stream_ptr()), "sampt_vos_index_masks")
        return out
    logits = logits.clone()                                                       # host-side helper for CPU tests
    if query_timestep is not None:
        for m in range(M):
            t = int(query_timestep[m])
            logits[m, :t] = -1e8
            if query_masks is not None:
                logits[m, t] = torch.where(query_masks[m] > 0, 1e8, -1e8)
    bg = torch.zeros((1, T, H, W), dtype=logits.dtype, device=logits.device)
    prob = torch.softmax(torch.cat([bg, logits], dim=0), dim=0)
    return prob.argmax(dim=0).to(torch.uint8)


def gather_masks(masks: torch.Tensor, max_frames: int, force: bool = False) -> Optional[torch.Tensor]:
    """All ranks call with their uint8 (T_local,H,W) masks; rank 0 receives (world, max_frames, H, W) (zero padded).
    Single fixed-shape ``gather`` to rank 0 over RCCL/xGMI (or gloo on CPU): every rank sends its own 0.59 MB/frame once —
    the payload SURVEY.md §8e states — instead of an all_gather's N copies.  ``force``: issue the collective even in a
    one-rank process group (the RCCL readiness test: the very same call path on hardware when only one GPU is available)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return masks[None]
    T, H, W = masks.shape
    pad = torch.zeros((max_frames, H, W), dtype=torch.uint8, device=masks.device)
    pad[:T] = masks
    if dist.get_rank() == 0:
        outs = [torch.empty_like(pad) for _ in range(dist.get_world_size())]
        dist.gather(pad, outs, dst=0)
        return torch.stack(outs)
    dist.gather(pad, None, dst=0)
    return None


def fnet_shard_usable(n_frames: int, world: int, batch: int) -> bool:
    """The pyramid all_gather of ``FnetShard`` is entered from inside ``model(...)``, which a rank without a frame batch never
    calls: the exchange is only safe when EVERY rank of the job owns at least one frame batch.  Pure function of (T, world,
    batch), so every rank reaches the same verdict without talking to the others."""
    return all(len(frame_batches(n_frames, world, k, batch)) > 0 for k in range(world))


def sharded_forward(model, video, batch: int = 8, shard_fnet: bool = False, emulate=None, force_collectives: bool = False):
    """One clip over all ranks (BASELINE config #5 / SURVEY.md §8e): frame batches of ``batch`` frames are dealt round
    robin (``frame_batches``); a rank runs the image encoder and the mask decoder on ITS frames only and contributes their
    uint8 index masks to one fixed-shape gather — with the default ``shard_fnet=False`` that gather is the ONLY collective,
    which is the mode north_star describes ("RCCL over xGMI only for the final mask gather"): every rank runs the tracker
    (encoder + window chain) on the whole clip.  ``shard_fnet=True`` is the opt-in extension: the tracker's per-frame encoder
    is frame-sharded too and the feature pyramid is all-gathered (``FnetShard``, 25 MB per 576x1024 frame — a real exchange
    step that north_star does not have); it is enabled only when every rank owns a frame batch (``fnet_shard_usable``: a rank
    without frames never enters ``model`` and would leave the others waiting in the all_gather).  The window chain —
    sequential in time by construction (pips/tracker.py:67-148), latency-bound, identical on every rank — is replicated.
    Returns (index masks (T,H,W) uint8 on rank 0 / None elsewhere, the rank's own SamPt output dict).

    ``force_collectives``: run the pyramid all_gather (with ``shard_fnet``) and the mask gather even in a ONE-rank process group
    (tests/test_gpu_dist_nccl.py: RCCL init + both collectives on hardware where only one GPU is available).

    ``emulate=(r, N)``: no process group — this process runs exactly rank r's share of an N-rank job, collectives stubbed
    (``FnetShard(emulate=True)``; the mask gather is skipped); the FnetShard is returned in the output dict's place of honour
    ``out["fnet_shard"]`` for the caller's clock corrections."""
    if emulate is not None:
        rank, world = int(emulate[0]), int(emulate[1])
    else:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    T = len(video["image"])
    batch = max(1, min(batch, -(-T // world)))          # short clips: shrink the batches so that most ranks get frames
    use_fs = shard_fnet and (world > 1 or force_collectives) and fnet_shard_usable(T, world, batch)   # the same verdict on every rank
    fs = FnetShard(rank, world, emulate=emulate is not None) if use_fs else None
    if fs is not None:
        video = {**video, "fnet_shard": fs}
    mine = [t for r in frame_batches(T, world, rank, batch) for t in r]
    if mine:
        out = model({**video, "frame_ids": mine})
        masks = index_masks(torch.stack(out["logits"], dim=0))                   # (len(mine), H, W) from (M, len(mine), H, W)
    else:                                               # more ranks than frame batches: this rank only joins the gather
        out = None
        masks = torch.zeros((0,) + tuple(video["target_hw"]), dtype=torch.uint8, device=model.device)
    if out is not None and fs is not None:
        out["fnet_shard"] = fs
    if emulate is not None:
        return masks, out
    per_rank = max(len([t for r in frame_batches(T, world, k, batch) for t in r]) for k in range(world))
    gathered = gather_masks(masks, per_rank, force=force_collectives)
    if rank != 0:
        return None, out
    full = torch.zeros((T,) + tuple(masks.shape[1:]), dtype=torch.uint8, device=masks.device)
    for k in range(world):
        ids = [t for r in frame_batches(T, world, k, batch) for t in r]
        full[torch.as_tensor(ids, dtype=torch.long, device=masks.device)] = gathered[k, :len(ids)]
    return full, out
