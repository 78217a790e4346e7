"""``SamPt`` — host orchestration of SAM-PT with the same constructor and ``forward(video)`` contract as the reference
(sam_pt/modeling/sam_pt.py:21-236), re-hosted for a device-resident pipeline.

The reference drives its predictor one call at a time with a host round trip after every call
(``.cpu().numpy()`` at sam_pt.py:830-831, numpy prompt assembly at :726-758, PIL/numpy frame hop at :849).  Here:

  * frames stay on the device; the whole clip is image-encoded in batches (``SamPredictor.encode_frames``) and the
    embeddings stay in HBM;
  * all prompts of the clip are assembled once on the host (the trajectories arrive on the CPU anyway —
    point_tracker/tracker.py:82-83) and uploaded in one copy;
  * each (frame, object) runs ``SamPredictor.track_decode``: the 1-2 prompt passes + R refinement passes +
    IoU-threshold rejection of ``predict_mask`` (sam_pt.py:760-837) as one device-side chain without host syncs.

With a predictor that lacks ``encode_frames``/``track_decode`` (e.g. the CPU oracle in the tests, or upstream's
``SamPredictor``) the same class falls back to the reference's call-by-call protocol (``set_image`` /
``predict_torch``), which is also what the unchanged reference ``SamPt`` does with our predictor.

Also built: ``query_masks`` mode with host-side query-point selection
(``sam_pt_amd/query_points.py``: random and k-medoids) and point re-initialisation with all four ``reinit_variant``s
(sam_pt.py:355-543), re-using the cached image embeddings across re-initialisation segments.
Shi-Tomasi / "mixed" point selection restate OpenCV's algorithms on the host (parity unpinned, query_points.py).
The optional patch-similarity filter (sam_pt.py:597-682, off in every shipped config) is built on the host with a
restated ``rgb2lab`` (skimage absent: that one function is parity unpinned).
"""
from __future__ import annotations

import os
import time
from enum import IntEnum
from typing import List, Optional

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


def rgb2lab(rgb_u8: torch.Tensor) -> torch.Tensor:
    """(...,3) uint8 sRGB -> CIELAB (D65, 2 degree observer), float64: the published algorithm of skimage.color.rgb2lab
    (sRGB companding, the sRGB->XYZ matrix, white (0.95047, 1, 1.08883), cube-root / linear f).  skimage is absent here:
    parity unpinned."""
    rgb = rgb_u8.double() / 255.0
    rgb = torch.where(rgb > 0.04045, ((rgb + 0.055) / 1.055) ** 2.4, rgb / 12.92)
    M = torch.tensor([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]],
                     dtype=torch.float64)
    xyz = rgb @ M.T / torch.tensor([0.95047, 1.0, 1.08883], dtype=torch.float64)
    f = torch.where(xyz > 0.008856, xyz.clamp(min=0) ** (1.0 / 3.0), 7.787 * xyz + 16.0 / 116.0)
    x, y, z = f[..., 0], f[..., 1], f[..., 2]
    return torch.stack([116.0 * y - 16.0, 500.0 * (x - y), 200.0 * (y - z)], dim=-1)


class PointVisibilityType(IntEnum):
    """Visibility codes stored as floats in ``visibilities`` (sam_pt/utils/util.py:267-282)."""
    VISIBLE = 1
    INVISIBLE = 0
    REINIT_FAILED = -1
    OUTSIDE_FRAME = -2
    PATCH_NON_SIMILAR = -3
    REJECTED_AFTER_PATCH_WAS_NON_SIMILAR = -4


def _stack_frames(frames):
    """torch.stack(frames), without the copy when the frames already ARE consecutive slices of one contiguous buffer (a decoded clip
    handed over frame by frame, as the reference's list-of-frames format does: sam_pt.py:150 stacks them again)."""
    f0 = frames[0]
    if isinstance(f0, torch.Tensor) and f0.is_contiguous() and len(frames) > 1:
        step = f0.numel() * f0.element_size()
        base = f0.data_ptr()
        if all(isinstance(f, torch.Tensor) and f.shape == f0.shape and f.dtype == f0.dtype and f.device == f0.device and f.is_contiguous()
               and f.data_ptr() == base + i * step and f.untyped_storage().data_ptr() == f0.untyped_storage().data_ptr()
               for i, f in enumerate(frames)):
            return torch.as_strided(f0, (len(frames),) + tuple(f0.shape), (f0.numel(),) + tuple(f0.stride()))
    return torch.stack(list(frames), dim=0)


class PendingForward:
    """Handle of a clip submitted with ``SamPt.forward_begin``; ``SamPt.forward_end`` turns it into ``forward``'s result."""

    def __init__(self, complete, result=None):
        self._complete, self._result = complete, result

    def result(self):
        if self._complete is not None:
            self._result, self._complete = self._complete(), None
        return self._result


class SamPt(nn.Module):
    def __init__(self, point_tracker, sam_predictor, sam_iou_threshold: float,
                 positive_point_selection_method: str = "kmedoids", negative_point_selection_method: str = "mixed",
                 positive_points_per_mask: int = 8, negative_points_per_mask: int = 0,
                 add_other_objects_positive_points_as_negative_points: bool = True,
                 max_other_objects_positive_points: Optional[int] = None, point_tracker_mask_batch_size: int = 5,
                 iterative_refinement_iterations: int = 12, use_patch_matching_filtering: bool = False,
                 patch_size: int = 3, patch_similarity_threshold: float = 0.01, use_point_reinit: bool = False,
                 reinit_point_tracker_horizon: int = 24, reinit_horizon: int = 24,
                 reinit_variant: str = "reinit-at-median-of-area-diff"):
        super().__init__()
        self.point_tracker = point_tracker
        self.sam_predictor = sam_predictor
        self.sam_iou_threshold = sam_iou_threshold
        self._sam = sam_predictor.model  # sam_pt.py:96 (makes .to(device) reach the Sam module when it is one)
        self.iterative_refinement_iterations = iterative_refinement_iterations
        self.positive_point_selection_method = positive_point_selection_method
        self.negative_point_selection_method = negative_point_selection_method
        self.positive_points_per_mask = positive_points_per_mask
        self.negative_points_per_mask = negative_points_per_mask
        self.add_other_objects_positive_points_as_negative_points = add_other_objects_positive_points_as_negative_points
        self.max_other_objects_positive_points = max_other_objects_positive_points
        self.point_tracker_mask_batch_size = point_tracker_mask_batch_size
        self.use_patch_matching_filtering = use_patch_matching_filtering
        self.patch_size, self.patch_similarity_threshold = patch_size, patch_similarity_threshold
        self.use_point_reinit = use_point_reinit
        self.reinit_point_tracker_horizon, self.reinit_horizon = reinit_point_tracker_horizon, reinit_horizon
        self.reinit_variant = reinit_variant
        self.profile = {}
        self.compute_unused_query_masks = False        # fused path: see forward()
        self.overlap_tracker_and_encoder = True        # fused path only: tracker on a second HIP stream (see forward)
        # persistent GEMM workgroups per XCD (of 32 CUs) for the encoder batches while the tracker runs beside them: an int or
        # one entry per batch (the last repeats); 0 = every CU; None (the default) = resolved per ViT precision in
        # ``_encoder_gemm_workgroups`` (28 for fp16, every CU for f16x3).  An explicit setting — attribute or
        # SAMPT_ENC_WGS="28,28,32" — is always honoured as given (experiments; that one —
        # every CU for the last batch, whose second half the tracker's rounds no longer share — measured 111.0 against 114.7 fps
        # with clips in flight: the decoder chain and the next clip's tracker encoder want those CUs, profiles/r4_c27_*).
        env = os.environ.get("SAMPT_ENC_WGS")
        # (an entry "30/28/30/28" gives one count per launch kind: qkv / proj / fc1 / fc2)
        self.encoder_gemm_workgroups_beside_tracker = ([tuple(int(x) for x in v.split("/")) if "/" in v else int(v)
                                                        for v in env.split(",")] if env else None)
        # fused path only.  The decoder chain always runs on a third (non-default, hence hipGraph-capturable) stream after
        # the encoder; pipeline_decoder=True instead starts the chains of each encoder batch as soon as that batch is done,
        # and overlap_tracker_encoder_fnet=True moves the tracker's own encoder to the side stream too.  Both were
        # measured on MI355X (profiles/r2_v2_*) and LOSE: the ViT GEMMs already fill the chip, so concurrent compute-bound
        # work only contends (fp16 GEMM 421 -> 469 us, tracker convs 2-3x longer) and 8-item decoder chains cost 1.7x the
        # GPU time of one 24-item chain: 77.9 fps serial, 75.2 with the tracker encoder overlapped, 71.9 with both.
        self.pipeline_decoder = False
        self.overlap_tracker_encoder_fnet = False
        # fused path only, round 6: split the chip IN SPACE while the tracker's window rounds run beside the encoder.  The side
        # streams (window rounds, decoder chains) are created confined to the last `side_cus_per_xcd` CUs of every XCD and the image
        # encoder runs on a stream confined to the others (hipExtStreamCreateWithCUMask through sampt_stream_create_cu_range), so a
        # round's launches never queue behind an attention or LayerNorm launch that covers the whole chip (measured round 5 / 6:
        # the chain takes 53 - 84 ms alone and 145 - 150 ms beside the encoder; its launches waited up to 1 ms for a CU) and
        # the encoder's kernels never find "their" CUs taken.  0 = off (priority streams + the GEMM workgroup knob, rounds 3 - 5).
        # SAMPT_SIDE_CUS overrides.
        self.side_cus_per_xcd = int(os.environ.get("SAMPT_SIDE_CUS", "0"))
        self._side_stream = None
        self._dec_stream = None
        self._enc_stream = None
        self._streams_split = None
        self.timeline = None      # measurement hook: a dict that forward() fills with timed CUDA events / host stamps

    @property
    def device(self):
        return self._sam.device

    # ------------------------------------------------------------------------------------------------
    def forward(self, video):
        return self._forward_guarded(video, defer=False)

    def _forward_guarded(self, video, defer):
        if self.training:
            raise NotImplementedError(f"{self._get_name()} does not support training...")
        from . import _lib, prefetch
        fused = hasattr(self.sam_predictor, "encode_frames") and hasattr(self.sam_predictor, "track_decode")
        try:
            with _lib.device_guard(self.device):      # streams / events / launches all on the model's device
                if fused:                             # the fused path holds the clip's embeddings itself
                    with prefetch.suspended():
                        return self._forward_impl(video, defer)
                return self._forward_impl(video, False)
        finally:                                      # never leave a clip's feature pyramid cached in the tracker
            if hasattr(self.point_tracker, "_prepared"):
                self.point_tracker._prepared = None
                self.point_tracker._prepared_events = None

    # -- clips in flight ---------------------------------------------------------------------------------------
    # ``forward`` ends with the decoder chain: ~20 ms of small, latency-bound launches that leave most of the GPU idle, and
    # it has to wait for them (the scores go back to the host).  A loop over clips (the reference's evaluator,
    # vos_eval/evaluator.py: one ``model(video)`` per sequence) gets that time back by submitting clip i + 1 — whose first
    # stage, the tracker's encoder, is throughput-bound — BEFORE collecting clip i: ``forward_begin`` enqueues everything
    # (the decoder chain on its own stream, after the encoder's event) and returns a handle, ``forward_end`` waits for that
    # clip's event only and builds the result.  ``forward(v)`` is ``forward_end(forward_begin(v))`` in effect: same kernels,
    # same streams, same results (tests/test_gpu_modules.py::test_stream_of_clips_equals_forward).
    def forward_begin(self, video) -> "PendingForward":
        r = self._forward_guarded(video, defer=True)
        return r if isinstance(r, PendingForward) else PendingForward(None, r)

    def forward_end(self, pending: "PendingForward"):
        from . import _lib
        with _lib.device_guard(self.device):
            return pending.result()

    def stream(self, videos):
        """Generator over ``forward`` results of an iterable of clips, one clip in flight ahead of the one being collected."""
        prev = None
        for v in videos:
            cur = self.forward_begin(v)
            if prev is not None:
                yield self.forward_end(prev)
            prev = cur
        if prev is not None:
            yield self.forward_end(prev)

    def _forward_impl(self, video, defer=False):
        images = _stack_frames(video["image"]) if isinstance(video["image"], (list, tuple)) else video["image"]
        n_frames, channels, height, width = images.shape
        assert images.dtype == torch.uint8, "Input images must be in uint8 format (0-255)"
        fused = hasattr(self.sam_predictor, "encode_frames") and hasattr(self.sam_predictor, "track_decode")
        # Frame sharding (sam_pt_amd.dist.sharded_forward; BASELINE config #5): ``video["frame_ids"]`` restricts the SAM
        # stage (encoder + decoder) to those frames; the tracker always sees the whole clip.  Logits then have
        # len(frame_ids) frames, trajectories / visibilities all of them.
        frame_ids = video.get("frame_ids")
        if frame_ids is not None and (self.use_point_reinit or not fused or video.get("query_masks") is not None):
            raise NotImplementedError("frame_ids needs the device path in query_points mode without re-initialisation")
        feats = None
        if fused:
            images = images.to(self.device)
        if video.get("query_masks") is None and video.get("query_points") is None:
            raise ValueError("No query points or masks provided")
        query_masks = None
        if video.get("query_masks") is not None:                         # VOS task (sam_pt.py:171-177)
            assert video.get("query_points") is None
            query_masks = video["query_masks"].float()
            query_points = self.extract_query_points(images, query_masks, video["query_point_timestep"])
        else:                                                            # demo (sam_pt.py:178-182)
            query_points = video["query_points"]
        tracked = None
        if fused:
            limit = getattr(self.sam_predictor, "max_prompt_points", None)
            if limit is not None:        # fail before the clip is encoded, not after (the reference accepts any prompt size)
                m_, p_, _ = query_points.shape
                others = (m_ - 1) * self.positive_points_per_mask if self.add_other_objects_positive_points_as_negative_points else 0
                if self.max_other_objects_positive_points is not None:
                    others = min(others, self.max_other_objects_positive_points)
                if p_ + others > limit:
                    raise ValueError(f"prompts of up to {p_ + others} points ({m_} objects x {p_} points, other objects' "
                                     f"positives as negatives) exceed the decoder's limit of {limit}: lower "
                                     "point_tracker_mask_batch_size / set max_other_objects_positive_points")
            # The image encoder (compute-bound, no host syncs) and the point tracker (many small launches, one host sync
            # per round) only share the input frames: the whole clip's encoder work is enqueued on the current stream and
            # the tracker then runs on a second, high-priority stream, filling the GPU around the big GEMMs.
            # Streams: the tracker's encoder (compute-bound) and then the SAM encoder batches go to the current stream, the
            # tracker's latency-bound window rounds (one host sync each) run beside the SAM encoder on a high-priority
            # side stream, and the decoder chain runs on a third stream once the trajectories are on the host and the
            # encoder is done (see the knobs in __init__ for the measured alternatives).
            overlap = (not self.use_point_reinit) and images.is_cuda and self.overlap_tracker_and_encoder
            pipeline = None
            self._mark("start")
            if overlap:
                ready = torch.cuda.Event()
                ready.record()                                               # frames valid on this stream
                if hasattr(self.point_tracker, "prepare") and not self.overlap_tracker_encoder_fnet:
                    fs = video.get("fnet_shard")          # frame-sharded tracker encoder + pyramid all_gather (dist.FnetShard)
                    self.point_tracker.to(self.device).prepare(images, **({"shard": fs} if fs is not None else {}))
                    if not getattr(self.point_tracker, "chunk_events_on_other_stream", False):
                        ready = torch.cuda.Event()       # tracker without per-chunk events: wait for the whole pyramid
                        ready.record()
                from . import _lib
                split = int(self.side_cus_per_xcd) if 0 < int(self.side_cus_per_xcd) < 32 else 0
                if self._side_stream is None or self._streams_split != split:
                    if split:
                        # (the decoder chains need the whole chip — on the window rounds' 32 CUs a clip's chain took 115 ms
                        #  instead of 22, profiles/r6_c4_* — so only the rounds and the encoder are confined)
                        self._side_stream = _lib.cu_range_stream(images.device, 32 - split, 32)
                        self._dec_stream = torch.cuda.Stream(device=images.device, priority=-1)
                        self._enc_stream = _lib.cu_range_stream(images.device, 0, 32 - split)
                    else:
                        self._side_stream = torch.cuda.Stream(device=images.device, priority=-1)
                        self._dec_stream = torch.cuda.Stream(device=images.device, priority=-1)
                        self._enc_stream = None
                    self._streams_split = split
            sam_images = images if frame_ids is None else images[torch.as_tensor(frame_ids, device=images.device)]
            batch_events = [] if overlap else None
            self._mark("prepared")
            # the tracker's window rounds run BESIDE the encoder on the side stream: leave them whole CUs (a persistent GEMM
            # workgroup owns its CU; measured on MI355X, profiles/r3_v3_timeline_wgs*.log: 32 / 30 workgroups per XCD
            # 248 ms per clip with the tracker ending 24 ms after the encoder, 28 per XCD 229 ms)
            reserve = self._encoder_gemm_workgroups(int(query_points.shape[0] * query_points.shape[1])) if overlap else None
            enc_stream = self._enc_stream if overlap else None
            if enc_stream is not None:                   # spatial split: one persistent GEMM workgroup per CU the encoder owns
                reserve = 32 - self._streams_split
            kw = {"gemm_workgroups": reserve} if reserve else {}
            if enc_stream is not None:
                enc_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(enc_stream):
                    feats = self.sam_predictor.encode_frames(sam_images, chw=True, batch_events=batch_events, **kw)
                    self._mark("encoded")
            else:
                feats = self.sam_predictor.encode_frames(sam_images, chw=True, batch_events=batch_events, **kw)   # embeddings in HBM
                self._mark("encoded")
            if overlap:
                self._side_stream.wait_event(ready)
                with torch.cuda.stream(self._side_stream):
                    if hasattr(self.point_tracker, "prepare") and self.overlap_tracker_encoder_fnet:
                        self.point_tracker.to(self.device).prepare(images)
                    tracked = self._track_points(images, query_points)
                    self._mark("tracked")
                torch.cuda.current_stream().wait_stream(self._side_stream)
                if enc_stream is not None:
                    torch.cuda.current_stream().wait_stream(enc_stream)
                    if hasattr(feats, "record_stream"):
                        feats.record_stream(torch.cuda.current_stream())
                if batch_events:
                    if not self.pipeline_decoder:                            # one chain for the clip, after the last batch
                        batch_events = [(batch_events[-1][0], batch_events[-1][1])]
                    elif isinstance(self.pipeline_decoder, int) and not isinstance(self.pipeline_decoder, bool) \
                            and 0 < self.pipeline_decoder < len(batch_events):
                        # two chains: the frames of the first `pipeline_decoder` encoder batches, then the rest
                        cut = self.pipeline_decoder
                        batch_events = [batch_events[cut - 1], batch_events[-1]]
                    pipeline = (batch_events, self._dec_stream)
        n_masks, n_points_per_mask, _ = query_points.shape
        if query_masks is None:
            # In query_points mode the reference computes the query masks (sam_pt.py:181) but only asserts their shape
            # (:186) and hands them to SuperGlue-style trackers (:189-191); with any other tracker they are dead work
            # (M extra encoder + decoder chains), so the device path skips them unless asked to (or a tracker wants them).
            if not fused or self.compute_unused_query_masks or hasattr(self.point_tracker, "set_masks"):
                query_masks = self.extract_query_masks(images, query_points, feats)
        assert query_masks is None or query_masks.shape == (n_masks, height, width)
        target_hw = tuple(video["target_hw"])

        def tail(trajectories, visibilities, logits, scores, scores_per_frame):
            resize_factor = torch.tensor(target_hw) / torch.tensor(logits.shape[-2:])
            assert (resize_factor[0] - resize_factor[1]).abs().item() < 0.01, "The resizing should have been isotropic"
            if tuple(logits.shape[-2:]) != target_hw:
                logits = self._resize_logits(logits, target_hw)
            trajectories = trajectories * resize_factor
            assert logits.shape == (n_masks, n_frames if frame_ids is None else len(frame_ids), target_hw[0], target_hw[1])
            assert trajectories.shape == (n_frames, n_masks, n_points_per_mask, 2)
            assert visibilities.shape == (n_frames, n_masks, n_points_per_mask)
            return {"logits": [m for m in logits], "scores": scores.tolist(), "scores_per_frame": scores_per_frame.tolist(),
                    "trajectories": trajectories, "visibilities": visibilities}

        if not self.use_point_reinit:
            trajectories, visibilities = tracked if tracked is not None else self._track_points(images, query_points)
            pl = pipeline if fused else None
            if frame_ids is None:
                sam_args = (images, trajectories, visibilities)
            else:
                ids = torch.as_tensor(frame_ids)
                sam_args = (sam_images, trajectories[ids], visibilities[ids])
            if defer and pl is not None:     # forward_begin: the decoder chain is enqueued, nobody waits for it yet
                pend = self._apply_sam_to_trajectories(*sam_args, feats, pl, defer=True)

                def complete():
                    _, logits, scores_per_frame = self._finish_sam_fused(pend)
                    self._mark("decoded")
                    return tail(trajectories, visibilities, logits, scores_per_frame.mean(dim=0), scores_per_frame)
                return PendingForward(complete)
            _, logits, scores_per_frame = self._apply_sam_to_trajectories(*sam_args, feats, pl)
            scores = scores_per_frame.mean(dim=0)
            self._mark("decoded")
        else:
            trajectories, visibilities, logits, scores, scores_per_frame = self._forward_w_reinit(images, query_points, feats)
        return tail(trajectories, visibilities, logits, scores, scores_per_frame)

    def _encoder_gemm_workgroups(self, n_chains: int = 8):
        """The knob ``encoder_gemm_workgroups_beside_tracker`` resolved: an explicit value (int, list, SAMPT_ENC_WGS) as given;
        the default None -> for the fp16 ViT 30 per XCD when the tracker runs up to 8 chains (its split-fp16 mixer launches then
        fit the two CUs per XCD that leaves: 207 - 211 ms per clip against 214 with 28, profiles/r6_c5_*, r6_c7_*, r6_c8_*), 28 for
        more chains or another tracker (rounds 3 - 5: profiles/r3_v3_timeline_wgs*.log); every CU for f16x3 (the split-fp16
        encoder takes twice as long, so the tracker's rounds end long before it either way: 59.5 fps against 58.5 with 28 per
        XCD, profiles/r4_c8_bench_x3_wgs*.log)."""
        v = self.encoder_gemm_workgroups_beside_tracker
        if v is not None:
            return v
        if getattr(getattr(self.sam_predictor, "model", None), "precision", None) == "f16x3":
            return None
        small = n_chains <= 8 and type(self.point_tracker).__name__ == "PipsPointTracker" and os.environ.get("SAMPT_PIPS_MIXER", "2") == "2"
        return 30 if small else 28

    def _mark(self, name):
        if self.timeline is not None and torch.cuda.is_available():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                                   # on the stream that is current where the mark sits
            self.timeline[name] = (ev, time.perf_counter())

    def extract_query_points(self, images, query_masks, query_points_timestep):
        """Query points (M, P+ + P-, 3) = (t, x, y) from masks: positives from the mask, negatives from its complement
        (sam_pt.py:238-288).  Host-side selection, see sam_pt_amd/query_points.py."""
        from .query_points import extract_query_points_xy
        query_masks = query_masks.cpu()
        query_points_timestep = query_points_timestep.cpu()
        # k-medoid clustering on the device when the frames live there (same points bit for bit; query_points.py)
        dev = images.device if isinstance(images, torch.Tensor) and images.is_cuda else None
        xy = extract_query_points_xy(images, query_masks, query_points_timestep, self.positive_point_selection_method,
                                     self.positive_points_per_mask, device=dev)
        if self.negative_points_per_mask > 0:
            neg = extract_query_points_xy(images, [1 - qm for qm in query_masks], query_points_timestep,
                                          self.negative_point_selection_method, self.negative_points_per_mask, device=dev)
            xy = [torch.cat(x, dim=0) for x in zip(xy, neg)]
        xy = torch.stack(xy, dim=0)
        t = query_points_timestep[:, None, None].repeat(1, xy.shape[1], 1)
        return torch.concat([t, xy], dim=2)

    # ------------------------------------------------------------------------------------------------
    # point re-initialisation (sam_pt.py:355-543).  The image embeddings of the clip are computed
    # once and re-used by every re-initialisation segment (the reference re-encodes frames each time).
    def _forward_w_reinit(self, images, query_points, feats=None):
        n_frames = images.shape[0]
        tr_r, vi_r, lg_r, _, sc_r = self._forward_w_reinit_inner(images, query_points, feats)
        qf = query_points.clone()
        qf[:, :, 0] = n_frames - query_points[:, :, 0] - 1
        tr_l, vi_l, lg_l, _, sc_l = self._forward_w_reinit_inner(images.flip(0), qf, feats.flip(0) if feats is not None else None)
        tr_l, vi_l, lg_l = tr_l.flip(0), vi_l.flip(0), lg_l.flip(1)
        # NOTE: like the reference (sam_pt.py:380-384) the scores of the flipped pass are NOT flipped back
        ts = query_points[:, 0, 0].int()
        trajectories = torch.full_like(tr_r, torch.nan)
        visibilities = torch.full_like(vi_r, False)
        logits = torch.full_like(lg_r, torch.nan)
        scores_per_frame = torch.full_like(sc_r, torch.nan)
        for m, t in enumerate(ts):
            trajectories[t:, m], trajectories[:t, m] = tr_r[t:, m], tr_l[:t, m]
            visibilities[t:, m], visibilities[:t, m] = vi_r[t:, m], vi_l[:t, m]
            logits[m, t:], logits[m, :t] = lg_r[m, t:], lg_l[m, :t]
            scores_per_frame[t:, m], scores_per_frame[:t, m] = sc_r[t:, m], sc_l[:t, m]
        assert not torch.isnan(trajectories).any()
        assert not torch.isnan(logits).any()
        return trajectories, visibilities, logits, scores_per_frame.nanmean(dim=0), scores_per_frame

    def _forward_w_reinit_inner(self, images, query_points, feats=None):
        n_frames, _, height, width = images.shape
        n_masks, points_per_mask, _ = query_points.shape
        assert self.reinit_point_tracker_horizon >= self.reinit_horizon
        H = self.reinit_horizon
        dev = images.device if feats is not None else torch.device("cpu")
        trajectories = torch.full((n_frames, n_masks, points_per_mask, 2), torch.nan, dtype=torch.float32)
        visibilities = torch.full((n_frames, n_masks, points_per_mask), False, dtype=torch.float32)
        scores_per_frame = torch.full((n_frames, n_masks), torch.nan, dtype=torch.float32)
        logits = torch.full((n_masks, n_frames, height, width), torch.nan, dtype=torch.float32, device=dev)
        current = query_points.clone()
        for start in range(int(query_points[:, 0, 0].int().min()), n_frames):
            end = min(start + H, n_frames)
            end_trk = min(start + self.reinit_point_tracker_horizon, n_frames)
            cur_t = current[:, 0, 0].int()
            tracked = cur_t == start
            if tracked.sum() == 0:
                continue
            q_i = current[tracked].clone()
            q_i[:, :, 0] -= start
            assert (q_i[:, :, 0] == 0).all()
            traj_i, vis_i = self._track_points(images[start:end_trk], q_i)
            traj_i, vis_i = traj_i[:H], vis_i[:H]
            _, logits_i, spf_i = self._apply_sam_to_trajectories(images[start:end], traj_i, vis_i,
                                                                 feats[start:end] if feats is not None else None)
            logits_i = logits_i.type(torch.float32)
            logits[tracked.to(logits.device), start:end] = logits_i
            pred = (logits_i > 0).cpu()
            trajectories[start:end, tracked] = traj_i
            visibilities[start:end, tracked] = vis_i
            scores_per_frame[start:end, tracked] = spf_i
            if end == n_frames:
                continue
            # choose the frame to re-initialise from (sam_pt.py:467-505)
            area = pred[:, 1:, :, :].sum([2, 3]).float()
            area[area <= 25] = torch.nan
            if H // 4 < area.shape[1]:
                area[:, :H // 4] = torch.nan
            n_i = pred.shape[0]
            if self.reinit_variant == "reinit-on-horizon-and-sync-masks":
                nxt = H - 1 - 1
                others = cur_t[cur_t > start]
                if len(others) > 0:
                    nxt = min(nxt, others.min() - start - 1)
                q_t = torch.full((n_i,), nxt, dtype=torch.int64)
            elif self.reinit_variant == "reinit-at-median-of-area-diff":
                q_t = area.nanmedian(dim=1).indices
            elif self.reinit_variant == "reinit-on-similar-mask-area":
                target = pred[:, 0, :, :].sum([1, 2])
                diff = torch.abs(area - target[:, None])
                diff[diff.isnan()] = torch.inf
                q_t = diff.argmin(dim=1)
            elif self.reinit_variant == "reinit-on-similar-mask-area-and-sync-masks":
                target = pred[:, 0, :, :].sum([1, 2])
                diff = torch.abs(area - target[:, None]) / target[:, None]
                diff[diff.isnan()] = 720
                per_frame = diff.sum(dim=0)
                others = cur_t[cur_t > start]
                if len(others) > 0:
                    per_frame[others.min().item() - start - 1] -= 36
                q_t = torch.full((n_i,), per_frame.argmin(dim=0), dtype=torch.int64)
            else:
                raise ValueError(f"Unknown reinit variant: {self.reinit_variant}")
            invalid = area[torch.arange(n_i), q_t] <= 0          # NaN area (tiny / empty mask) compares False, as upstream
            if (~invalid).sum() > 0:
                q_masks = pred[:, 1:, :, :][torch.arange(n_i), q_t].type(torch.float32)
                upd = self.extract_query_points(images[start + 1:end], q_masks[~invalid], q_t[~invalid])
                valid_tracked = tracked.clone()
                valid_tracked[tracked] = ~invalid
                current[valid_tracked] = upd.to(current.device)
                current[valid_tracked, :, 0] += start + 1
            if invalid.sum() > 0:
                invalid_tracked = tracked.clone()
                invalid_tracked[tracked] = invalid
                current[invalid_tracked, :, 0] = n_frames
                current[invalid_tracked, :, 1:] = 0
                trajectories[end:, invalid_tracked] = -72
                visibilities[end:, tracked] = PointVisibilityType.REINIT_FAILED.value
                logits[invalid_tracked.to(logits.device), end:] = -float("inf")
        return trajectories, visibilities, logits, scores_per_frame.nanmean(dim=1), scores_per_frame

    @staticmethod
    def _resize_logits(logits, target_hw):
        from . import _lib
        with _lib.device_guard(logits.device):
            return SamPt._resize_logits_impl(logits, target_hw)

    @staticmethod
    def _resize_logits_impl(logits, target_hw):
        """Bilinear (align_corners=False) resize of the (M,T,H,W) logits to target_hw (sam_pt.py:205-206)."""
        if not logits.is_cuda:
            return F.interpolate(logits, size=target_hw, mode="bilinear", align_corners=False)   # reference protocol on CPU
        from . import _lib
        lib = _lib.load()
        M, T, H, W = logits.shape
        out = torch.empty((M, T, target_hw[0], target_hw[1]), dtype=torch.float32, device=logits.device)
        _lib.check(lib.sampt_resize_logits(_lib.ptr(logits.contiguous()), M * T, H, W, _lib.ptr(out), target_hw[0],
                                           target_hw[1], _lib.stream_ptr()), "sampt_resize_logits")
        return out

    # ------------------------------------------------------------------------------------------------
    def extract_query_masks(self, images, query_points, feats=None):
        """SAM on each object's query frame with its own query points (sam_pt.py:308-335).  The reference encodes the
        query frame once per object even when objects share it (App. B-3); with cached embeddings that is a lookup."""
        frame_ids = [int(t.item()) for t in query_points[:, 0, 0]]
        sub_feats = feats[frame_ids] if feats is not None else None
        _, logits, _ = self._apply_sam_to_trajectories(
            images=torch.stack([images[i] for i in frame_ids], dim=0),
            trajectories=query_points[:, None, :, 1:].cpu(),
            visibilities=torch.ones_like(query_points[:, None, :, 0]).cpu(),
            feats=sub_feats)
        return (logits > self.sam_predictor.model.mask_threshold)[0]

    def _track_points(self, rgbs, query_points):
        """Chunks of ``point_tracker_mask_batch_size`` objects per tracker call (sam_pt.py:545-576, 578-692)."""
        trajs, viss = [], []
        n_masks = query_points.shape[0]
        rgbs_dev = rgbs.to(self.device).unsqueeze(0)
        self.point_tracker.eval()
        for i in range(0, n_masks, self.point_tracker_mask_batch_size):
            q = query_points[i:i + self.point_tracker_mask_batch_size]
            m, p, _ = q.shape
            with torch.no_grad():
                out = self.point_tracker.to(self.device).evaluate_batch(rgbs_dev, q.reshape(1, m * p, 3).to(self.device))
            t = out["trajectories_pred"].squeeze(0)
            v = out["visibilities_pred"].squeeze(0).float()
            if self.use_patch_matching_filtering:
                v = self._patch_similarity_filter(rgbs, q.reshape(m * p, 3).cpu(), t, v, m, p)
            h, w = rgbs.shape[-2:]
            v[t[:, :, 0] / w < 0.01] = PointVisibilityType.OUTSIDE_FRAME.value   # sam_pt.py:684-690
            v[t[:, :, 1] / h < 0.01] = PointVisibilityType.OUTSIDE_FRAME.value
            v[t[:, :, 0] / w > 0.99] = PointVisibilityType.OUTSIDE_FRAME.value
            v[t[:, :, 1] / h > 0.99] = PointVisibilityType.OUTSIDE_FRAME.value
            trajs.append(t.reshape(-1, m, p, 2))
            viss.append(v.reshape(-1, m, p))
        return torch.cat(trajs, dim=1), torch.cat(viss, dim=1)

    # ------------------------------------------------------------------------------------------------
    def _patch_similarity_filter(self, rgbs, query_points, traj, vis, n_masks, n_pts):
        """Optional patch-similarity filter (sam_pt.py:597-682; off in every shipped config): a point whose CIELAB patch
        differs too much from its query patch is marked PATCH_NON_SIMILAR and everything beyond it (away from the query
        frame) REJECTED_AFTER_PATCH_WAS_NON_SIMILAR.  Host-side like the reference; ``rgb2lab`` restates
        ``skimage.color.rgb2lab`` (absent here, parity unpinned) — the rest is pinned against the reference with that
        function substituted.  rgbs (T,3,H,W) uint8, query_points (N,3), traj (T,N,2), vis (T,N) float."""
        rgbs = rgbs.cpu()
        lab = rgb2lab(rgbs[:, [2, 1, 0], :, :].permute(0, 2, 3, 1)).permute(0, 3, 1, 2).float()   # (channel swap: :645)
        ps = self.patch_size

        def patches(lab_frames, xy):                                                # (F,3,h,w), (F,K,2) -> (F,K,ps*ps,3)
            _, _, h, w = lab_frames.shape
            tmpl = torch.arange(-(ps // 2), ps // 2 + 1)
            tmpl = torch.stack(torch.meshgrid(tmpl, tmpl, indexing="ij"), dim=-1).reshape(-1, 2)
            grid = ((xy[:, :, None, :] + tmpl[None, None] + 0.5) / torch.tensor([w, h])[None, None, None]) * 2 - 1
            return F.grid_sample(lab_frames, grid.float(), align_corners=False, mode="bilinear").permute(0, 2, 3, 1)

        qt = query_points[:, 0].long()
        qp = patches(lab[qt], query_points[:, None, 1:].float()).squeeze(1)         # (N,ps*ps,3)
        tp = patches(lab, traj.cpu().float())                                       # (T,N,ps*ps,3)
        diff = tp.flatten(2, 3) - qp[None].flatten(2, 3)
        similar = torch.exp(-torch.norm(diff, dim=-1) / (2 * ps ** 2)) > self.patch_similarity_threshold
        vis = vis.clone()
        vis[(vis == 1) & ~similar] = PointVisibilityType.PATCH_NON_SIMILAR.value
        T = vis.shape[0]
        for i in range(n_masks * n_pts):
            t0 = int(query_points[i, 0].item())
            for t in range(t0 + 1, T):
                if vis[t, i] == PointVisibilityType.PATCH_NON_SIMILAR.value:
                    vis[t + 1:, i] = PointVisibilityType.REJECTED_AFTER_PATCH_WAS_NON_SIMILAR.value
                    break
            for t in range(t0 - 1, -1, -1):
                if vis[t, i] == PointVisibilityType.PATCH_NON_SIMILAR.value:
                    vis[:t, i] = PointVisibilityType.REJECTED_AFTER_PATCH_WAS_NON_SIMILAR.value
                    break
        return vis

    def _prepare_points(self, trajectories, visibilities, frame_idx, mask_idx, n_masks):
        """Prompt assembly of sam_pt.py:726-758: visible points (== 1), tail points negative, the other objects'
        visible positives appended as negatives.  ``trajectories`` / ``visibilities``: torch tensors, or — the fused path
        converts once per clip, per-item torch indexing was 0.1 ms per (frame, object) — numpy arrays (T, M, P, 2) float and
        (T, M, P) bool."""
        if isinstance(trajectories, torch.Tensor):
            trajectories, visibilities = trajectories.numpy(), (visibilities == 1).numpy()
        point_coords = trajectories[frame_idx, mask_idx]
        point_labels = np.ones((len(point_coords)), dtype=int)
        if self.negative_points_per_mask > 0:
            point_labels[self.positive_points_per_mask:] = 0
        vmask = visibilities[frame_idx, mask_idx]
        coords = point_coords[vmask]
        labels = point_labels[vmask]
        if n_masks > 1 and self.add_other_objects_positive_points_as_negative_points:
            pp = self.positive_points_per_mask
            keep = np.arange(n_masks) != mask_idx
            others = trajectories[frame_idx, keep, :pp][visibilities[frame_idx, keep, :pp]]   # object-major, like the cat
            if self.max_other_objects_positive_points is not None and len(others) > self.max_other_objects_positive_points:
                idx = np.random.choice(len(others), self.max_other_objects_positive_points, replace=False)
                others = others[idx, :]
            coords = np.concatenate([coords, others], axis=0)
            labels = np.concatenate([labels, np.zeros((len(others)), dtype=int)], axis=0)
        return coords, labels

    def _apply_sam_to_trajectories(self, images, trajectories, visibilities, feats=None, pipeline=None, defer=False):
        n_frames, channels, height, width = images.shape
        _, n_masks, points_per_mask, _ = trajectories.shape
        assert trajectories.shape == (n_frames, n_masks, points_per_mask, 2)
        assert visibilities.shape == (n_frames, n_masks, points_per_mask)
        trajectories, visibilities = trajectories.cpu(), visibilities.cpu()
        if feats is not None:
            if pipeline is None:
                return self._apply_sam_fused(images, trajectories, visibilities, feats)
            events, stream = pipeline
            with torch.cuda.stream(stream):       # everything of the SAM stage is allocated and launched on `stream`
                pend = self._enqueue_sam_fused(images, trajectories, visibilities, feats, events)
                pend["event"] = torch.cuda.Event()
                pend["event"].record()            # this clip's chain (and the copy of its scores to the host) ends here
            return pend if defer else self._finish_sam_fused(pend)
        return self._apply_sam_stepwise(images, trajectories, visibilities)

    # -- device-resident path --------------------------------------------------------------------------------
    def _apply_sam_fused(self, images, trajectories, visibilities, feats, batch_events=None):
        return self._finish_sam_fused(self._enqueue_sam_fused(images, trajectories, visibilities, feats, batch_events))

    def _finish_sam_fused(self, pend):
        """Second half of the SAM stage: wait for the chain enqueued by ``_enqueue_sam_fused`` (its own event when it ran on a
        decoder stream — later clips' chains on that stream are not waited for), hand the logits over to the caller's
        stream, bring the scores home: the only sync of the SAM stage."""
        logits, scores = pend["logits"], pend["scores"]
        if pend.get("event") is not None:
            pend["event"].synchronize()
            cur = torch.cuda.current_stream()
            cur.wait_event(pend["event"])
            logits.record_stream(cur)             # the logits are consumed on the caller's stream from here on
        elif scores.is_cuda:
            torch.cuda.current_stream().synchronize()
        scores_cpu = (pend["host"].clone() if pend["host"] is not None else scores.cpu()).view(pend["n_frames"], pend["n_masks"])
        return self._mean_scores(scores_cpu), logits, scores_cpu

    def _enqueue_sam_fused(self, images, trajectories, visibilities, feats, batch_events=None):
        """First half: prompts assembled on the host, every decoder chain of the clip enqueued on the current stream.
        ``batch_events``: [(end_frame, event)] of the encoder batches (``SamPredictor.encode_frames``).  When given, the
        current stream is a decoder stream that must not wait for the whole encoder: the items are chunked per encoder
        batch and every chunk waits only for the event of the batch that holds its frames."""
        n_frames, _, height, width = images.shape
        n_masks = trajectories.shape[1]
        dev = self.device
        pred = self.sam_predictor
        size = (height, width)
        prompts = []
        kmax = 1
        traj_np, vis_np = trajectories.numpy(), (visibilities == 1).numpy()
        for t in range(n_frames):
            for m in range(n_masks):
                c, l = self._prepare_points(traj_np, vis_np, t, m, n_masks)
                if len(c):
                    c = pred.transform.apply_coords(c, size)
                prompts.append((c, l))
                kmax = max(kmax, len(c))
        use_graph = bool(getattr(pred, "use_graph", False)) and hasattr(pred, "decode_staging") and images.is_cuda
        if use_graph:        # point slots in steps of 8: clips whose longest prompts differ by a point share their staging
            kmax = -(-kmax // 8) * 8     # buckets and captured graphs (the slots past an item's count are never read)
        xy = np.zeros((len(prompts), kmax, 2), dtype=np.float32)
        lab = np.zeros((len(prompts), kmax), dtype=np.int32)
        for i, (c, l) in enumerate(prompts):
            xy[i, :len(c)] = c
            lab[i, :len(c)] = l
        # All (frame, object) items with a non-empty prompt run as ONE batched device-side chain per chunk of Fmax items,
        # whatever their number of visible points: the batch is ragged (per-item point counts, padding tokens masked in
        # the decoder's attention), so an item's result equals its un-batched one.
        items = [i for i, (c, l) in enumerate(prompts) if len(c) > 0]            # empty prompt: sam_pt.py:766-767
        k_all = np.array([len(c) for c, _ in prompts], dtype=np.int32)
        npos_all = np.array([int((l == 1).sum()) for _, l in prompts], dtype=np.int32)
        two_pass = self.negative_points_per_mask > 0
        Fmax = getattr(pred.model, "max_decode_batch", 1)
        chunks = []                                                               # [(items, event to wait for or None)]
        if batch_events:
            lo = 0
            for end, ev in batch_events:
                group = [i for i in items if lo <= i // n_masks < end]
                chunks += [(group[s0:s0 + Fmax], ev) for s0 in range(0, len(group), Fmax)]
                lo = end
        else:
            chunks = [(items[s0:s0 + Fmax], None) for s0 in range(0, len(items), Fmax)]
        cur_stream = torch.cuda.current_stream() if batch_events else None
        lib = None
        if use_graph:
            from . import _lib
            lib = _lib.load()
            # every output element is written by exactly one item; only items WITHOUT a prompt keep the -inf the reference gives them
            logits = torch.empty((n_masks, n_frames, height, width), dtype=torch.float32, device=dev)
            scores = torch.empty((n_frames * n_masks,), dtype=torch.float32, device=dev)
            if len(items) < len(prompts):
                _lib.check(lib.sampt_fill_f32(_lib.ptr(logits), logits.numel(), -float("inf"), _lib.stream_ptr()), "sampt_fill_f32")
                _lib.check(lib.sampt_fill_f32(_lib.ptr(scores), scores.numel(), -float("inf"), _lib.stream_ptr()), "sampt_fill_f32")
            # A chunk's prompts are packed on the host, in chunk order, straight into a pinned mirror of its bucket's device block and
            # cross in ONE asynchronous copy (enqueued with the chunk below: from pinned memory the host never waits for the stream,
            # so a clip submitted with forward_begin is not held up behind the encoder's event).
            staged, used = [], {}
            for chunk, _ in chunks:
                st = pred.decode_staging(len(chunk), kmax, size)
                k = used.get(id(st), 0)                    # the k-th chunk of this clip in this bucket: its own pinned mirror
                used[id(st)] = k + 1
                mir = st["mirror"](k)
                if mir["free"] is not None:
                    mir["free"].synchronize()              # the previous clip's copy out of this mirror (long done; never waits)
                ch = np.asarray(chunk, dtype=np.int64)
                mir["pts"][...] = xy[ch]
                mir["labels"][...] = lab[ch]
                mir["k"][...] = k_all[ch]
                mir["npos"][...] = npos_all[ch]
                mir["items"][...] = ch
                staged.append((st, mir))
        else:
            xy_d = torch.from_numpy(xy).to(dev)
            lab_d = torch.from_numpy(lab).to(dev)
            k_d, npos_d = torch.from_numpy(k_all).to(dev), torch.from_numpy(npos_all).to(dev)
            logits = torch.full((n_masks, n_frames, height, width), -float("inf"), dtype=torch.float32, device=dev)
            scores = torch.full((n_frames * n_masks,), -float("inf"), dtype=torch.float32, device=dev)
            idx_d = [torch.tensor(chunk, dtype=torch.long, device=dev) for chunk, _ in chunks]
        for ci, (chunk, ev) in enumerate(chunks):
            if ev is not None:
                cur_stream.wait_event(ev)
            F_ = len(chunk)
            ks, ps = k_all[chunk], npos_all[chunk]
            ragged = bool((ks != ks[0]).any() or (two_pass and (ps != ps[0]).any()))
            if use_graph:
                # a replayed hipGraph has its pointers baked in: the chunk's embeddings are gathered into the bucket's
                # persistent buffers and its outputs land there (SamPredictor.decode_staging), by item number
                st, mir = staged[ci]
                # (the bucket's device block is shared by the chunks that use it: filled here, in stream order, from the chunk's own
                #  pinned mirror — an asynchronous copy, the host does not wait for the encoder)
                st["block"].copy_(mir["host"], non_blocking=True)
                mir["free"] = torch.cuda.Event()
                mir["free"].record()
                emb, hq = (feats.emb, feats.hq) if hasattr(feats, "emb") else (feats, None)
                sp, P = _lib.stream_ptr(), _lib.ptr
                _lib.check(lib.sampt_move_rows(P(emb), P(st["feats"]), P(st["items"]), F_, emb[0].numel() * 4, n_masks, n_frames, 0, sp),
                           "sampt_move_rows")
                f_in = st["feats"]
                if hq is not None:
                    _lib.check(lib.sampt_move_rows(P(hq), P(st["hq"]), P(st["items"]), F_, hq[0].numel() * 4, n_masks, n_frames, 0, sp),
                               "sampt_move_rows")
                    f_in = type(feats)(st["feats"], st["hq"])
                out_l, out_s = st["logits"], st["score"]
                pred.track_decode(f_in, st["pts"], st["labels"], int(ks.max()), int(ps.max()) if two_pass else -1,
                                  int(self.iterative_refinement_iterations), float(self.sam_iou_threshold), size, out_l,
                                  out_s, k_item=st["k_item"] if ragged else None,
                                  npos_item=st["npos_item"] if (ragged and two_pass) else None, graph=True)
                _lib.check(lib.sampt_move_rows(P(out_l), P(logits), P(st["items"]), F_, height * width * 4, n_masks, n_frames, 1, sp),
                           "sampt_move_rows")
                _lib.check(lib.sampt_move_rows(P(out_s), P(scores), P(st["items"]), F_, 4, 1, 1, 1, sp), "sampt_move_rows")
            else:
                idx = idx_d[ci]
                t_idx, m_idx = idx // n_masks, idx % n_masks
                out_l = torch.empty((F_, height, width), dtype=torch.float32, device=dev)
                out_s = torch.empty((F_,), dtype=torch.float32, device=dev)
                pred.track_decode(feats.index_select(0, t_idx), xy_d.index_select(0, idx).contiguous(),
                                  lab_d.index_select(0, idx).contiguous(), int(ks.max()), int(ps.max()) if two_pass else -1,
                                  int(self.iterative_refinement_iterations), float(self.sam_iou_threshold), size,
                                  out_l, out_s, k_item=k_d.index_select(0, idx).contiguous() if ragged else None,
                                  npos_item=npos_d.index_select(0, idx).contiguous() if (ragged and two_pass) else None)
                logits[m_idx, t_idx] = out_l
                scores[idx] = out_s
        host = None
        if scores.is_cuda:                        # asynchronous copy into pinned memory: nobody waits here
            host = torch.empty(scores.shape, dtype=scores.dtype, pin_memory=True)
            host.copy_(scores, non_blocking=True)
        return {"logits": logits, "scores": scores, "host": host, "n_frames": n_frames, "n_masks": n_masks, "event": None,
                "feats": feats}                    # (the embeddings stay referenced until the chain that reads them is done)

    @staticmethod
    def _mean_scores(scores_per_frame):
        valid = torch.isfinite(scores_per_frame)
        cnt = valid.sum(0).clamp(min=1)
        return (torch.where(valid, scores_per_frame, torch.zeros_like(scores_per_frame)).sum(0) / cnt).numpy()

    # -- reference protocol (any SamPredictor-compatible object) ---------------------------------------------
    def _apply_sam_stepwise(self, images, trajectories, visibilities):
        n_frames, _, height, width = images.shape
        n_masks = trajectories.shape[1]
        pred = self.sam_predictor
        dev = self.device
        logits = torch.full((n_masks, n_frames, height, width), -float("inf"), dtype=torch.float32)
        scores = torch.full((n_frames, n_masks), -float("inf"), dtype=torch.float32)
        for t in range(n_frames):
            pred.set_image(images[t].permute(1, 2, 0).cpu().numpy())
            for m in range(n_masks):
                c, l = self._prepare_points(trajectories, visibilities, t, m, n_masks)
                if len(c) == 0:
                    continue
                pc = torch.as_tensor(pred.transform.apply_coords(c, pred.original_size), dtype=torch.float, device=dev)
                pl = torch.as_tensor(l, dtype=torch.int, device=dev)
                kw = dict(boxes=None, multimask_output=False, return_logits=True)
                if self.negative_points_per_mask == 0:
                    ml, iou, low = pred.predict_torch(point_coords=pc[None], point_labels=pl[None], mask_input=None, **kw)
                else:
                    _, _, low = pred.predict_torch(point_coords=pc[pl == 1][None], point_labels=pl[pl == 1][None],
                                                   mask_input=None, **kw)
                    ml, iou, low = pred.predict_torch(point_coords=pc[None], point_labels=pl[None], mask_input=low, **kw)
                for _ in range(int(self.iterative_refinement_iterations)):
                    msk = ml[0, 0] > 0
                    if msk.sum() < 2:
                        break
                    yx = msk.nonzero()
                    box = torch.tensor([yx[:, 1].min(), yx[:, 0].min(), yx[:, 1].max(), yx[:, 0].max()],
                                       dtype=torch.float, device=dev)
                    ml, iou, low = pred.predict_torch(point_coords=pc[None], point_labels=pl[None], boxes=box[None, None, :],
                                                      mask_input=low, multimask_output=False, return_logits=True)
                sc = float(iou[0, 0])
                scores[t, m] = sc
                if sc >= self.sam_iou_threshold:
                    logits[m, t] = ml[0, 0].float().cpu()
        return self._mean_scores(scores), logits, scores
