#!/usr/bin/env python
"""bench.py — end-to-end SAM-PT frames/sec on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one synthetic clip: ``SamPt.forward(video)`` (PIPS tracking of 8 query
points, SAM ViT image encoding of every frame, 1 + R prompt/refinement decoder passes per frame) plus the
background-stack softmax/argmax of the reference's timed window (sam_pt/vos_eval/eval.py:262-268, 304-337).
Frames are uint8 tensors already resident in HBM when the timed region starts.  With N > 1 every rank processes
its own clip (sequence sharding, no data-path collective) and the final uint8 masks are gathered to rank 0 over
RCCL inside the timed region; value = all frames of all ranks / max-over-ranks time.

``value`` is the metric as SURVEY.md §8(d) defines it: one BLOCKING ``SamPt.forward`` per step (what the reference evaluator's
loop does with the drop-in).  The timed loop alternates TWO distinct seeded clips (same geometry and query points, different
textures) and asserts that every step encoded its clip (``clips_encoded == steps``): nothing can be reused across steps.
``value_pipelined`` is the same K steps once more with one clip kept in flight ahead of the one being collected
(``SamPt.forward_begin`` / ``forward_end``: the decoder chain of clip i overlaps the tracker encoder of clip i + 1; exactly
``--steps`` clips are submitted AND collected between the two barriers); ``--submit pipelined`` makes that the ``value``.

Launch: ``python bench.py --gpus 1`` or ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="vit_h", choices=["vit_h", "vit_l", "vit_b"])
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--points", type=int, default=8)
    ap.add_argument("--objects", type=int, default=1, help="number of tracked objects (BASELINE config #4 uses 3)")
    ap.add_argument("--refine", type=int, default=12)
    ap.add_argument("--precision", default="f16", choices=["f16", "f16x3", "f32"],
                    help="ViT block arithmetic: f16 = fp16 MFMA inputs (the headline mode); f16x3 = every product from split-fp16 "
                         "pieces (3 fp16 MFMAs, fp32 accumulate): the reference's fp32 arithmetic at fp32 grade; f32 = exact f32 MFMA")
    ap.add_argument("--encode-batch", type=int, default=8)
    ap.add_argument("--decode-batch", type=int, default=128, help="max (frame, object) items per batched decoder chain")
    ap.add_argument("--tracker", default="pips", choices=["pips", "pips_plus_plus", "cotracker"],
                    help="point tracker (the metric is quoted on PIPS; CoTracker = BASELINE configs #3/#5, row a13; "
                         "PIPS++ = SURVEY.md §8 row f4)")
    ap.add_argument("--neg-points", type=int, default=0, help="negative query points per object (config #3: 8 + 8)")
    ap.add_argument("--square", type=int, default=0, help="square synthetic frames of this size (config #5: 1024)")
    ap.add_argument("--dec-pipeline", action="store_true",
                    help="start the decoder chains of each encoder batch as soon as that batch is done (measured: loses)")
    ap.add_argument("--dec-split", type=int, default=0,
                    help="two decoder chains: frames of the first N encoder batches as soon as they are encoded, then the rest")
    ap.add_argument("--side-cus", type=int, default=None,
                    help="CUs per XCD reserved for the side streams (tracker window rounds, decoder chains); the image encoder runs on "
                         "a stream confined to the other CUs (SamPt.side_cus_per_xcd; 0 = priority streams only)")
    ap.add_argument("--overlap-fnet", action="store_true", help="tracker encoder on the side stream too (measured: loses)")
    ap.add_argument("--no-dec-graph", action="store_true", help="decode chains as plain launches instead of hipGraph replays")
    ap.add_argument("--shard", default="sequences", choices=["sequences", "frames", "lpt"],
                    help="N > 1: 'sequences' = one clip per rank (weak scaling, the default the metric uses); 'frames' = ONE "
                         "clip, its frame batches dealt over the ranks (strong scaling, BASELINE config #5); 'lpt' = a "
                         "DAVIS-2017-val-like set of --sequences clips (34-104 frames) LPT-assigned to the ranks (strong "
                         "scaling of BASELINE config #4, reports the load imbalance)")
    ap.add_argument("--emulate-ranks", default="",
                    help="one GPU stands in for every rank r of an N-rank --shard frames job in turn (comma-separated N, e.g. "
                         "2,4,8): runs exactly rank r's share (its frames of the tracker encoder, the replicated window chain, its "
                         "frame batches of the SAM stage), collectives stubbed; reports the predicted per-rank time and speed-up "
                         "(`frame_sharding_model` in the JSON line)")
    ap.add_argument("--allgather-gbs", type=float, default=250.0,
                    help="--emulate-ranks: assumed RCCL all_gather rate per GPU over xGMI, GB/s of received bytes (7 links x 153 "
                         "GB/s peak per GPU; ring collectives are per-link bound)")
    ap.add_argument("--lpt-model", default="",
                    help="one GPU: time ONE blocking forward per distinct sequence length of the DAVIS-17-val histogram (--sequences), "
                         "then predict the N-rank time of a sequence-sharded pass (comma-separated N, e.g. 2,4,8) from those MEASURED "
                         "per-sequence times with an LPT assignment by time (`sequence_sharding_model` in the JSON line; a model: the "
                         "ranks' host threads and the final mask gather are not in it)")
    ap.add_argument("--sequences", type=int, default=30, help="--shard lpt: number of sequences of the DAVIS-17 val histogram")
    ap.add_argument("--native-480p", action="store_true",
                    help="feed the 480x854 frames as they are (tracker at 480p, SAM resizes inside) instead of the "
                         "reference pipelines' pre-resize to 576x1024")
    ap.add_argument("--hq", action="store_true", help="HQ-SAM decoder (reference default samhq_vit_huge; BASELINE config #5)")
    ap.add_argument("--pips-vis-bias", type=float, default=2.0,
                    help="bias of the random PIPS visibility head (weights.py default 2.0 -> sigmoid 0.88, just under the 0.9 "
                         "link threshold: short hops, ~23 tracker rounds per clip; 4.0 behaves like a trained model on "
                         "trackable points: 7-frame hops).  Sensitivity knob only; the headline number uses the default.")
    ap.add_argument("--cotracker-delta-scale", type=float, default=None,
                    help="scale of the random CoTracker flow head (weights.init_cotracker_state_dict; default 0.003).  Over the 15 "
                         "chained windows of a 64-frame clip the default makes the ORACLE ITSELF move by 0.4 px under a 1e-7 weight "
                         "perturbation (DESIGN.md section 2): long-clip parity lines use 0.001")
    ap.add_argument("--fnet-exact", action="store_true",
                    help="tracker encoder convolutions as exact fp32 MFMAs instead of the 3-term split-fp16 MFMAs")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (cpu_baseline + parity)")
    ap.add_argument("--parity-frames", type=int, default=24,
                    help="frames of the clip whose SAM stage the CPU oracle repeats (evenly spaced; default: every frame of the "
                         "24-frame clip, ~10 s of host time per ViT-H frame)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (fp32 ViT, query-mask pass, IoU 0.7)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--submit", default="sequential", choices=["pipelined", "sequential"],
                    help="what `value` times.  sequential (default): one blocking SamPt.forward per step — the metric as SURVEY.md "
                         "section 8(d) defines it; the pipelined loop is timed afterwards as `value_pipelined`.  pipelined: the step loop "
                         "keeps ONE clip in flight ahead of the one it collects (SamPt.forward_begin / forward_end: the decoder chain of "
                         "clip i overlaps the tracker encoder of clip i + 1; exactly --steps clips are submitted AND collected inside "
                         "the timed region) and the blocking loop is reported as `value_per_forward`")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the second timed loop (value_pipelined / value_per_forward)")
    ap.add_argument("--shard-fnet", action="store_true",
                    help="--shard frames: also shard the tracker's per-frame encoder and all_gather its feature pyramid (an extra "
                         "collective north_star does not have; default off: the uint8 mask gather is the only collective)")
    return ap.parse_args()


def build_model(args, dev):
    if args.fnet_exact:
        os.environ["SAMPT_FNET_F16X3"] = "0"
    from sam_pt_amd.point_tracker import PipsPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    sam = SamHip(args.model, precision=args.precision, seed=72, max_batch=args.encode_batch, hq=args.hq,
                 max_decode_batch=args.decode_batch).to(dev)
    from sam_pt_amd.weights import init_pips_state_dict
    if args.tracker == "pips":
        tracker = PipsPointTracker(state_dict=init_pips_state_dict(72, vis_bias=args.pips_vis_bias), fnet_chunk=8)
    elif args.tracker == "cotracker":
        from sam_pt_amd.point_tracker import CoTrackerPointTracker
        from sam_pt_amd.weights import init_cotracker_state_dict
        ckw = {} if args.cotracker_delta_scale is None else {"delta_scale": args.cotracker_delta_scale}
        tracker = CoTrackerPointTracker(state_dict=init_cotracker_state_dict(72, **ckw), fnet_chunk=8)   # configs/model/point_tracker/cotracker.yaml
    else:
        from sam_pt_amd.point_tracker import PipsPlusPlusPointTracker
        tracker = PipsPlusPlusPointTracker(seed=72, fnet_chunk=8)
    pred = SamPredictor(sam)
    if args.no_dec_graph:
        pred.use_graph = False
    model = SamPt(tracker, pred, **sampt_kwargs(args)).eval()
    model.pipeline_decoder = args.dec_split if args.dec_split else args.dec_pipeline
    model.overlap_tracker_encoder_fnet = args.overlap_fnet
    if args.side_cus is not None:
        model.side_cus_per_xcd = args.side_cus
    return model


def consume(model, out, max_frames):
    """What the reference's timed window does with a clip's result (eval.py:304-326) + the mask gather of a multi-GPU job."""
    from sam_pt_amd.dist import gather_masks, index_masks
    logits = torch.stack(out["logits"], dim=0)          # (M,T,H,W) on device
    if not logits.is_cuda:                               # the reference protocol returns host tensors (sam_pt.py:862-864)
        logits = logits.to(model.device)
    masks = index_masks(logits)                          # bg stack + softmax + argmax (eval.py:304-326)
    gathered = gather_masks(masks, max_frames)           # RCCL gather of uint8 masks (no-op for 1 GPU)
    return masks, gathered


def one_step(model, video, max_frames, shard="sequences", shard_fnet=False):
    from sam_pt_amd.dist import sharded_forward
    if shard == "frames":
        full, _ = sharded_forward(model, video, batch=8, shard_fnet=shard_fnet)   # rank 0: the (T,H,W) index masks; others: None
        return full, full
    return consume(model, model(video), max_frames)


class ClipsInFlight:
    """The pipelined step loop: ``submit`` enqueues a clip (SamPt.forward_begin) and THEN collects the previous one
    (forward_end + consume); ``flush`` collects the last.  K submits + one flush = K clips submitted and collected."""

    def __init__(self, model, max_frames):
        self.model, self.max_frames, self.pending, self.last = model, max_frames, None, None

    def submit(self, video):
        h = self.model.forward_begin(video)
        self.flush()
        self.pending = h
        return self.last

    def flush(self):
        if self.pending is not None:
            out, self.pending = self.model.forward_end(self.pending), None
            self.last = consume(self.model, out, self.max_frames)[0]
        return self.last


def gemm_roofline(args, dev, insitu=None, frame_hw=(576, 1024)):
    """Dominant kernel = the fp16 MFMA GEMM of the ViT encoder (gemm_f16_p8: 256 x 256 x 64 tiles, 8-phase LDS-DMA pipeline,
    persistent workgroups; csrc/gemm_f16_p8.hip).  ``achieved`` / ``avg_launch_us`` are IN SITU: one extra (untimed) step
    of the very same workload runs with every GEMM launch of the encoder bracketed by HIP events on its launching stream
    (sampt_vit_profile_begin/end), so the figure is the real launches' algorithmic FLOP / their summed duration, tracker
    overlap included, and matches the rocprofv3 average of profiles/.  ``isolated_*`` repeats each distinct launch (shape
    AND epilogue: bias, GELU, in-place residual) alone on an idle GPU, weighted by its count per encode call."""
    from sam_pt_amd import _lib
    from sam_pt_amd.weights import SAM_CONFIGS
    lib = _lib.load()
    cfg = SAM_CONFIGS[args.model]
    B, D = args.encode_batch, cfg.embed_dim
    Mg = B * cfg.grid * cfg.grid
    # blocks before the first global one run on the rows that hold pixels (VitEngine::live_rows), the others on the full grid;
    # windowed blocks multiply the real tokens only (padding rows are never multiplied)
    g0 = min(cfg.global_attn_indexes) if cfg.global_attn_indexes else cfg.depth
    h_tok = -(-frame_hw[0] // cfg.patch_size)
    lh = min(cfg.grid, -(-h_tok // cfg.window_size) * cfg.window_size) if frame_hw[1] == cfg.img_size else cfg.grid
    Ml = B * lh * cfg.grid
    n_live = g0 if lh < cfg.grid else 0
    shapes = []   # (M, N, K, dtype(2 = f16 out, 1 = f32 out), act, in-place residual, count per encode call)
    for M, cnt in ((Ml, n_live), (Mg, cfg.depth - n_live)):
        if cnt:
            shapes += [(M, 3 * D, D, 2, 0, 0, cnt), (M, D, D, 1, 0, 1, cnt), (M, 4 * D, D, 2, 2, 0, cnt), (M, D, 4 * D, 1, 0, 1, cnt)]
    tot_flop = tot_t = 0.0
    launches = 0
    # L2-miss (HBM + Infinity Cache) bytes per launch from the committed --pmc passes (tools/gemm_traffic.py); bench.py
    # cannot run rocprofv3 on itself, so the figure is looked up per shape and is null for shapes that were not profiled
    traffic_tab, tot_traffic, tot_alg_bytes = {}, 0.0, 0.0
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    x3 = args.precision == "f16x3"
    # PMC passes of the shipped kernel (tools/gemm_traffic.py over tools/gemm_bench.py): the newest committed file wins
    cands = [f for f in (["r5_gemm_hbm_traffic_x3.json"] if x3 else ["r5_gemm_hbm_traffic.json", "r3_gemm_hbm_traffic.json"])
             if os.path.exists(os.path.join(pdir, f))]
    tpath = os.path.join(pdir, cands[0]) if cands else os.path.join(pdir, "(none)")
    if cands:
        with open(tpath) as fh:
            traffic_tab = json.load(fh)["per_launch"]
    g = torch.Generator(device="cpu").manual_seed(0)
    for (M, N, K, dt, act, res, cnt) in shapes:
        A = torch.randn(M, K, generator=g) * 0.5
        W = torch.randn(N, K, generator=g) / K ** 0.5
        if x3:           # x3 rows in (hi | lo per 32 k), x3 rows (dtype 4) or f32 (dtype 3) out, alpha undoes the 2^8 weight scale
            from sam_pt_amd.pack import F16X3_WSHIFT, x3_rows
            A, W = x3_rows(A).to(dev), x3_rows(W, F16X3_WSHIFT).to(dev)
        else:
            A, W = A.half().to(dev), W.half().to(dev)
        bias = torch.zeros(N, device=dev)
        Cc = torch.zeros(M, 2 * N if (x3 and dt == 2) else N, device=dev, dtype=torch.float16 if dt == 2 else torch.float32)
        gdt, alpha = ((4 if dt == 2 else 3), 2.0 ** -8) if x3 else (dt, 1.0)
        call = lambda: lib.sampt_gemm_ex(gdt, _lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(Cc) if res else None, _lib.ptr(Cc),
                                         M, N, K, act, alpha, None, None, 0, 0, _lib.stream_ptr())
        for _ in range(12):                      # steady state (the first launches of a shape run 5 - 10 % slower)
            _lib.check(call(), "gemm")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        tot_flop += 2.0 * M * N * K * cnt
        tot_t += t * cnt
        launches += cnt
        tr = traffic_tab.get(f"{M},{N},{K},{dt}")
        if tr is None or tot_traffic is None:
            tot_traffic = None
        else:
            tot_traffic += (tr["read_bytes"] + tr["write_bytes"]) * cnt
            tot_alg_bytes += (tr["algorithmic_read_bytes"] + tr["algorithmic_write_bytes"]) * cnt
        del A, W, Cc
    iso = tot_flop / tot_t / 1e12
    ach, avg_us, n_launch = iso, tot_t / launches * 1e6, launches
    if insitu is not None and insitu[2] > 0:
        ach, avg_us, n_launch = insitu[0] / (insitu[1] * 1e-3) / 1e12, insitu[1] * 1e3 / insitu[2], insitu[2]
    # f16x3: every product costs three fp16 MFMAs, so the fp32-equivalent ceiling is a third of the dense fp16 peak
    peak = 2500.0 / 3 if x3 else 2500.0
    return {"bound": "mfma", "kernel": ("gemm_f16_p8<X3> (ViT qkv / proj / MLP GEMMs on x3 rows: hi.hi + hi.lo + lo.hi, 24 MFMAs per phase)"
                                       if x3 else "gemm_f16_p8 (ViT qkv / proj / MLP GEMMs: 256x256x64 tiles, 8 waves, 8-phase LDS-DMA fp16 "
                                       "MFMA pipeline, persistent workgroups)"),
            "achieved": round(ach, 1), "peak": round(peak, 1),
            "unit": "TFLOP/s fp32-equivalent (2 M N K / t; 3 fp16 MFMAs per product: peak = 2500 / 3)" if x3 else "TFLOP/s",
            "frac": round(ach / peak, 4),
            **({"mfma_work_TFLOPs": round(3 * ach, 1), "mfma_work_frac_of_dense_fp16_peak": round(3 * ach / 2500.0, 4)} if x3 else {}),
            "measured": "in situ: HIP events around every GEMM launch of one extra step" if insitu else "isolated shapes",
            "launches_timed": n_launch, "isolated_achieved": round(iso, 1),
            "isolated_avg_launch_us": round(tot_t / launches * 1e6, 1),
            "traffic": None if tot_traffic is None else round(tot_traffic / launches),
            "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE; counts Infinity-Cache hits)",
            "traffic_source": ("committed --pmc passes of the shipped kernel (" + os.path.basename(tpath) + "), looked up per shape — "
                               "not measured by this run (bench.py cannot run rocprofv3 on itself)") if cands else
                              "no --pmc passes are committed for this arithmetic (traffic: null)",
            "algorithmic_bytes_per_launch": None if tot_traffic is None else round(tot_alg_bytes / launches),
            "launches_per_encode_call": launches,
            "avg_launch_us": round(avg_us, 1), "encode_batch": B}


def secondary_rooflines(args, dev):
    """The other kernels north_star names, each alone on an idle GPU through its kernel-level C-ABI entry point at the
    shape the timed workload launches it with (HIP events, 10 launches): the fused correlation sampler against HBM
    bandwidth (algorithmic bytes: SURVEY.md §8d, 33.5 KB per (frame, point, level)), both flash-attention kernels and a
    tracker-encoder convolution against the dense fp16 MFMA peak (the 3-term split issues 3 MFMAs per product, so its
    fp32-equivalent ceiling is a third of 2.5 PF)."""
    from sam_pt_amd import _lib
    from sam_pt_amd.pack import split_f16x3
    from sam_pt_amd.weights import SAM_CONFIGS
    lib = _lib.load()
    cfg = SAM_CONFIGS[args.model]
    g = torch.Generator(device="cpu").manual_seed(1)

    def timed(fn, reps=40):            # the GEMM block's steady-state protocol: 15 launches to warm up, 40 timed
        for _ in range(15):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    out = []
    # correlation sampler: n = 2 x points chains (both directions batched), S = 8 frames, 4 levels
    n, S_ = 2 * args.points * args.objects, 8
    H0, W0 = 144, 256
    pyr = [torch.randn(S_, H0 >> l, W0 >> l, 128, generator=g).to(dev) for l in range(4)]
    fidx = torch.arange(S_, dtype=torch.int32).repeat(n, 1).contiguous().to(dev)
    ff = torch.randn(n, S_, 128, generator=g).to(dev)
    co = (torch.rand(S_, n, 2, generator=g) * torch.tensor([W0 - 20.0, H0 - 20.0]) + 10).to(dev)
    xo = torch.empty(n, S_, 196, device=dev)
    t = timed(lambda: lib.sampt_corr_sample_f32(_lib.ptr_array(pyr), H0, W0, _lib.ptr(fidx), S_, n, _lib.ptr(ff), _lib.ptr(co),
                                                _lib.ptr(xo), _lib.stream_ptr()))
    nbytes = S_ * n * 4 * (8 * 8 * 128 * 4 + 512 + 196)
    out.append({"kernel": "k_pips_corr_sample (fused correlation + 7x7 sampler)", "bound": "hbm", "achieved": round(nbytes / t / 1e9, 1),
                "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / t / 8e12, 4), "launch_us": round(t * 1e6, 2),
                "algorithmic_bytes_per_launch": nbytes, "note": f"{S_ * n * 4} wave-sized units per launch: launch-latency bound at this size"})
    # flash attention, global and windowed blocks (encode batch frames)
    heads, hd, D = cfg.num_heads, cfg.head_dim, cfg.embed_dim
    for name, B, S2 in (("global (64x64 tokens)", args.encode_batch, 64), ("windowed (14x14 tokens)", args.encode_batch * 25, 14)):
        N = S2 * S2
        qkv = (torch.randn(B * N, 3 * D, generator=g) * 0.5).half().to(dev)
        rh = (torch.randn(2 * S2 - 1, hd, generator=g) * 0.1).to(dev)
        rw = (torch.randn(2 * S2 - 1, hd, generator=g) * 0.1).to(dev)
        ao = torch.empty(B * N, D, dtype=torch.float16, device=dev)
        t = timed(lambda: lib.sampt_vit_attention_f16(_lib.ptr(qkv), _lib.ptr(rh), _lib.ptr(rw), _lib.ptr(ao), B, S2, heads, hd,
                                                      None, 0, _lib.stream_ptr()))
        fl = 4.0 * B * heads * N * N * hd
        ent = {"kernel": f"k_flash_f16, {name}", "bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": 2500.0,
               "unit": "TFLOP/s", "frac": round(fl / t / 2.5e15, 4), "launch_us": round(t * 1e6, 1)}
        if S2 == 14:
            # 4 N^2 hd FLOP over (q + k + v + out) = 4 N hd fp16 values per (window, head): N / 2 = 98 FLOP per byte against a
            # machine balance of ~312 — the windowed launch is HBM-bound by construction; its roofline is the q / k / v / out bytes
            nb = 4.0 * B * N * D * 2
            ent.update({"bound": "hbm", "achieved": round(nb / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nb / t / 8e12, 4),
                        "algorithmic_bytes_per_launch": int(nb),
                        "mfma": {"achieved": round(fl / t / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(fl / t / 2.5e15, 4)}})
        out.append(ent)
        del qkv, ao
    # tracker encoder: the 64 -> 64 3x3 convolution at half resolution (the most frequent fnet layer), 3-term split-fp16 with
    # pre-split activation planes (what the encoder's InstanceNorm hands it): the LDS-DMA kernel k_conv_f16x3_dma<64>
    nimg, Hc, Wc, ci, cc = 8, 288, 512, 64, 64
    x = torch.relu(torch.randn(nimg, Hc, Wc, ci, generator=g)).to(dev)
    xh = x.half()
    xhl = torch.stack([xh, (x - xh.float()).half()]).contiguous()
    w = torch.randn(cc, 9 * ci, generator=g) * (2.0 / (cc * 9)) ** 0.5
    whl, b = split_f16x3(w).to(dev), torch.zeros(cc, device=dev)
    y = torch.empty(nimg, Hc, Wc, cc, device=dev)
    t = timed(lambda: lib.sampt_conv2d_nhwc(4, _lib.ptr(xhl), _lib.ptr(whl), _lib.ptr(b), _lib.ptr(y), nimg, Hc, Wc, ci, cc, 3, 3, 1, 1,
                                            _lib.stream_ptr()), reps=20)
    fl = 2.0 * nimg * Hc * Wc * cc * 9 * ci
    out.append({"kernel": "k_conv_f16x3_dma<64> (fnet 64->64 3x3 @288x512, 8 frames, pre-split fp16 planes by LDS-DMA)", "bound": "mfma",
                "achieved": round(fl / t / 1e12, 1), "peak": 833.3,
                "unit": "TFLOP/s fp32-equivalent (3 fp16 MFMAs per product: 2500 / 3)", "frac": round(fl / t / 833.3e12, 4),
                "launch_us": round(t * 1e6, 1)})
    del x, xh, xhl, y
    # mask decoder, token -> image attention of one pass over the clip's (frame, object) items: K and V of every item are
    # read once (algorithmic bytes = 2 x items x 4096 x 128 x 4), queries / outputs are a few KB
    import ctypes as C
    items, nq = args.frames * args.objects, 7 + args.points + args.neg_points
    q = torch.randn(items, nq, 128, generator=g).to(dev)
    kk = torch.randn(items, 4096, 128, generator=g).to(dev)
    vv = torch.randn(items, 4096, 128, generator=g).to(dev)
    oo = torch.empty(items, nq, 128, device=dev)
    nb = C.c_size_t()
    _lib.check(lib.sampt_attention_t2i_workspace_bytes(items, nq, 4096, C.byref(nb)), "t2i workspace")
    wsb = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    t = timed(lambda: lib.sampt_attention_t2i_f32(_lib.ptr(q), _lib.ptr(kk), _lib.ptr(vv), _lib.ptr(oo), items, nq, 4096,
                                                  _lib.ptr(wsb), wsb.numel(), _lib.stream_ptr()))
    nbytes = 2 * items * 4096 * 128 * 4
    out.append({"kernel": f"k_attn_t2i_part + _merge (decoder token->image attention, {items} items x {nq} tokens)", "bound": "hbm",
                "achieved": round(nbytes / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / t / 8e12, 4),
                "launch_us": round(t * 1e6, 1), "algorithmic_bytes_per_launch": nbytes})
    return out


def sampt_kwargs(args, **over):
    kw = dict(sam_iou_threshold=-1e9, positive_points_per_mask=args.points, negative_points_per_mask=args.neg_points,
              iterative_refinement_iterations=args.refine, point_tracker_mask_batch_size=5)
    kw.update(over)
    return kw


def cpu_reference(args, frames, qp, out):
    """``cpu_baseline`` and ``parity`` from ONE pass of the CPU oracle over the timed workload (rank 0, N = 1).

    The oracle runs the REFERENCE ALGORITHM, not our optimised schedule: the whole clip through the PIPS oracle with fnet
    recomputed for every 8-frame window and the 6-iteration init pass (pips/tracker.py:42-153), then — on a bounded
    sample of frames, a ViT-H pass costs seconds — ``set_image`` + the 1 + R sequential ``predict_torch`` calls of
    ``SamPt.predict_mask`` (sam_pt.py:760-837, 848-858).  cpu fps = T / (tracker seconds + T x mean SAM seconds per
    sampled frame).  The same oracle outputs are the parity reference for the device result ``out`` of the timed
    configuration: trajectories in index space, visibilities, per-frame mask IoU (bar 1 - 1e-3)."""
    from oracle.parity import compare, reference_run
    from sam_pt_amd.weights import SAM_CONFIGS, init_pips_state_dict, init_sam_state_dict
    cores = min(os.cpu_count() or 1, 32)   # PyTorch-CPU collapses when oversubscribed on 256-thread hosts
    cfg = SAM_CONFIGS[args.model]
    sd = init_sam_state_dict(cfg, 72, hq=args.hq)
    T = frames.shape[0]
    n_par = max(1, min(args.parity_frames, T))
    ids = sorted({int(round(i * (T - 1) / max(n_par - 1, 1))) for i in range(n_par)}) if n_par > 1 else [T // 2]
    factory = None
    psd = None
    if args.tracker == "pips":
        psd = init_pips_state_dict(72, vis_bias=args.pips_vis_bias)
    elif args.tracker == "pips_plus_plus":
        from oracle.pips2_ref import Pips2TrackerRef
        from sam_pt_amd.weights import init_pips2_state_dict
        p2 = init_pips2_state_dict(72)
        factory = lambda: Pips2TrackerRef(p2)
    else:
        from oracle.cotracker_ref import CoTrackerTrackerRef
        from sam_pt_amd.weights import init_cotracker_state_dict
        csd = init_cotracker_state_dict(72, **({} if args.cotracker_delta_scale is None else {"delta_scale": args.cotracker_delta_scale}))
        factory = lambda: CoTrackerTrackerRef(csd)
    ref = reference_run(cfg, sd, psd, frames.cpu(), qp, sampt_kwargs(args), frame_ids=ids, hq=args.hq,
                        reference_cost=True, threads=cores, tracker_factory=factory)
    sec = ref["seconds"]
    sam_pf = (sec["encoder"] + sec["decoder"]) / len(ids)
    total = sec["tracker"] + T * sam_pf
    cpu = {"value": round(T / total, 4), "unit": "frames/s", "cores": cores, "kind": "port",
           "algorithm": "reference schedule (tracker encoder per window + init pass; call-by-call decoder)",
           "sample": f"PyTorch-CPU fp32 oracle, {cores} threads: point tracker over the WHOLE {T}-frame clip "
                     f"({sec['windows']} windows) {sec['tracker']:.1f}s measured; SAM stage on {len(ids)} of {T} frames "
                     f"{ids}: image encoder {sec['encoder'] / len(ids):.1f}s + {sec['predict_calls'] // len(ids)} "
                     f"predict_torch calls {sec['decoder'] / len(ids):.2f}s per frame, extrapolated to {T} frames"}
    par = compare(out, ref)
    par["precision"] = args.precision
    par["oracle_driver"] = ref.get("driver")
    # `pass` is the STRICT statement for every tracker: masks within 1e-3 IoU, trajectories identical after round(), visibilities
    # and rejections identical — on the workload this line timed.  For CoTracker two more figures say how to read a strict
    # failure on a long clip: `pass_off_boundary` ignores coordinates the ORACLE puts within 2e-3 px of an x.5 rounding boundary
    # (oracle/parity.py), and `oracle_noise_floor` is the oracle's distance to ITSELF under a 1e-7 relative perturbation of its
    # weights (oracle/noise_floor.py): with random weights and >= 12 chained windows that floor is tenths of a pixel (DESIGN.md
    # section 2), and `within_oracle_noise` says whether the device result is as close to the oracle as the oracle is to itself.
    par["bar"] = "mask IoU >= 1 - 1e-3 per frame; trajectories identical after round(); visibilities identical"
    common = bool(par["mask_iou_min"] >= 1 - 1e-3 and par["vis_identical"] and par["rejections_identical"])
    par["pass"] = bool(common and par["traj_index_identical"])
    if args.tracker == "cotracker":
        from oracle.noise_floor import tracker_noise_floor
        par["pass_off_boundary"] = bool(common and par["traj_index_identical_off_boundary"])
        nf = tracker_noise_floor(lambda s_: CoTrackerTrackerRef(s_), csd, frames.cpu()[None], qp.reshape(1, -1, 3))
        par["oracle_noise_floor"] = {"perturbation": "weights x (1 + 1e-7 N(0,1))", **nf["floor"]}
        par["within_oracle_noise"] = bool(par["traj_max_abs_px"] <= max(3.0 * nf["floor"]["traj_max_abs_px"], 5e-3))
    return cpu, par


def cached_parity(args, frames, qp, out):
    """``parity`` without the timing leg (``--no-cpu-baseline``): if oracle/make_cache.py left the oracle's result for this very
    workload under tests/golden/oracle_cache/ (oracle/cache.py: exact masks / trajectories / rejections), compare with it;
    None if there is no such file.  Development runs only — the default run computes the oracle live (``cpu_reference``)."""
    if args.tracker != "pips" or args.hq or args.square or args.neg_points or args.native_480p or args.pips_vis_bias != 2.0:
        return None
    from oracle import workloads as W
    from oracle.cache import load
    from oracle.parity import compare
    if args.objects != 1 or args.points != 8:
        return None
    w = W.bench_workload(args.model, args.frames)
    if not (torch.equal(w["frames"], frames) and torch.equal(w["qp"], qp) and w["kw"] == sampt_kwargs(args)):
        return None
    ref = load(W.key_of(w))
    if ref is None:
        return None
    par = compare(out, ref)
    par["precision"] = args.precision
    par["oracle_driver"] = ref.get("driver")
    par["pass"] = bool(par["mask_iou_min"] >= 1 - 1e-3 and par["traj_index_identical"] and par["vis_identical"]
                       and par["rejections_identical"])
    return par


def quick_fps(model, video, frames_n, steps=2):
    one_step(model, video, frames_n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step(model, video, frames_n)
    torch.cuda.synchronize()
    return round(frames_n * steps / (time.perf_counter() - t0), 2)


class ReferenceApiPredictor:
    """Hides everything the upstream ``segment_anything.SamPredictor`` does not have: ``SamPt`` then runs the REFERENCE
    protocol over the two HIP seams — tracker first, then per frame ``set_image(numpy frame)`` and 1-2 + R sequential
    ``predict_torch`` calls with a host round trip each (sam_pt/modeling/sam_pt.py:760-837, 848-858) — which is what the
    unchanged reference ``SamPt`` does with ``model.sam_predictor._target_=sam_pt_amd.sam_predictor.SamPredictor``."""
    _HIDDEN = ("encode_frames", "track_decode", "decode_staging", "set_features")

    def __init__(self, pred):
        object.__setattr__(self, "_p", pred)

    def __getattr__(self, name):
        if name in ReferenceApiPredictor._HIDDEN:
            raise AttributeError(name)
        return getattr(self._p, name)

    def __setattr__(self, name, value):
        setattr(self._p, name, value)


def reference_protocol_lines(args, model, video):
    """fps of the reference's own loop over the HIP seams, with and without the clip-embedding prefetch (prefetch.py)."""
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd import prefetch
    ref_model = SamPt(model.point_tracker, ReferenceApiPredictor(model.sam_predictor), **sampt_kwargs(args)).eval()
    n0 = prefetch.stats["clips_encoded"]
    res = {"reference_sampt_over_hip_seams": quick_fps(ref_model, video, args.frames, steps=1)}
    # every pass (warm-up and timed) encoded its clip: the timed step includes the ViT (a cache across passes would skip it)
    assert prefetch.stats["clips_encoded"] - n0 == 2, prefetch.stats
    os.environ["SAMPT_PREFETCH"] = "0"
    try:
        res["reference_sampt_over_hip_seams_no_prefetch"] = quick_fps(ref_model, video, args.frames, steps=1)
    finally:
        del os.environ["SAMPT_PREFETCH"]
    return res


def secondary_lines(args, model, video, dev):
    """Variants of the headline workload the judge asked to see beside it (2 timed steps each, same clip, every one with a
    blocking ``SamPt.forward`` per step — compare with the line's ``value_per_forward``): the split-fp16 and the exact-fp32
    ViT, the reference's dead query-mask SAM pass switched back on (sam_pt.py:181), the shipped IoU threshold 0.7, and the
    REFERENCE protocol (tracker, then set_image + sequential predict_torch per frame) over the two HIP seams — the speed a
    user of the unchanged reference ``SamPt`` gets from the two ``_target_`` overrides alone (1 timed step each)."""
    res = {}
    model.compute_unused_query_masks = True
    res["with_reference_query_mask_pass"] = quick_fps(model, video, args.frames)
    model.compute_unused_query_masks = False
    model.sam_iou_threshold = 0.7
    res["sam_iou_threshold_0.7"] = quick_fps(model, video, args.frames)
    model.sam_iou_threshold = -1e9
    # the mode every VOS run uses (sam_pt.py:171-177): query MASKS at t = 0 -> k-medoid query points (clustering on the device,
    # csrc/kmedoids.hip) -> the same forward
    from sam_pt_amd.synth import bench_query_masks
    qm = bench_query_masks(T=args.frames, seed=72, n_objects=args.objects, native=args.native_480p, square=args.square)
    video_qm = {k: v for k, v in video.items() if k != "query_points"}
    video_qm.update(query_masks=qm, query_point_timestep=torch.zeros(args.objects))
    res["query_masks_mode"] = quick_fps(model, video_qm, args.frames)
    res.update(reference_protocol_lines(args, model, video))
    if args.precision == "f16":
        import copy
        for prec in ("f16x3", "f32"):     # the reference's fp32 arithmetic: on the fp16 pipe at fp32 grade / on the f32 MFMA
            a2 = copy.copy(args)
            a2.precision = prec
            m2 = build_model(a2, dev)
            res["vit_precision_" + prec] = quick_fps(m2, video, args.frames)
            del m2
            torch.cuda.empty_cache()
    return {"unit": "frames/s", "steps": 2, **res}


def frame_sharding_model(args, model, video, steps=3):
    """Predicted in-clip scaling (``--shard frames``: dist.sharded_forward) from ONE GPU: for every N in --emulate-ranks and
    every rank r < N the GPU runs exactly rank r's share — tracker encoder on its frame share, the window chain (replicated),
    SAM encoder + decoder on its frame batches — with the collectives stubbed: the pyramid all_gather is replaced by computing
    the other ranks' frames between two events (that GPU time is subtracted) and priced at ``--allgather-gbs``; the final mask
    gather (0.59 MB per frame) is priced the same way.  predicted step = max over ranks; speed-up against the same clip's
    blocking forward on this GPU."""
    from sam_pt_amd.dist import sharded_forward
    T = len(video["image"])
    H, W = video["target_hw"]

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    base = timed(lambda: one_step(model, video, T))
    res = {"one_gpu_ms_per_clip": round(base, 2), "assumed_allgather_GBps": args.allgather_gbs, "steps": steps, "by_world": {}}
    for N in [int(v) for v in args.emulate_ranks.split(",") if v]:
        ranks = []
        for r in range(N):
            stub = {"ms": 0.0, "bytes": 0}

            def step():
                _, out = sharded_forward(model, video, batch=8, shard_fnet=True, emulate=(r, N))
                fs = out.get("fnet_shard") if out else None
                if fs is not None:
                    stub["ms"] += fs.stub_ms()
                    stub["bytes"] = fs.bytes_received
            step()
            torch.cuda.synchronize()
            stub["ms"] = 0.0
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps * 1e3
            own = wall - stub["ms"] / steps
            comm = (stub["bytes"] + T * H * W) / (args.allgather_gbs * 1e9) * 1e3     # pyramid all_gather + uint8 mask gather
            ranks.append({"rank": r, "own_ms": round(own, 2), "stubbed_ms": round(stub["ms"] / steps, 2),
                          "comm_ms_model": round(comm, 2), "predicted_ms": round(own + comm, 2)})
        worst = max(x["predicted_ms"] for x in ranks)
        res["by_world"][str(N)] = {"predicted_ms_per_clip": worst, "predicted_speedup": round(base / worst, 2), "ranks": ranks}
    return res


def sequence_sharding_model(args, model, dev, steps=2):
    """Predicted sequence-sharded scaling (BASELINE config #4: one DAVIS sequence per rank, north_star's >= 6 x at 8 GPUs) from
    ONE GPU: every distinct length of the DAVIS-2017-val histogram is timed as a blocking SamPt.forward of a synthetic clip of that
    length (so the fixed per-clip head and tail — tracker encoder start-up, window chain, decoder chain — are IN the numbers), the
    30 sequences are LPT-assigned to N ranks by measured time, predicted pass time = the slowest rank's sum.  A MODEL: per-rank host
    threads, PCIe and the uint8 mask gather (0.59 MB per frame) are not in it; no N > 1 run exists on this pool."""
    from sam_pt_amd.dist import DAVIS17_VAL_LENGTHS, lpt_assign
    from sam_pt_amd.synth import bench_clip
    lengths = DAVIS17_VAL_LENGTHS[:args.sequences]
    frames, qp = bench_clip(T=max(lengths), seed=72, n_pos=args.points, n_objects=args.objects, native=args.native_480p,
                            n_neg=args.neg_points, square=args.square)
    fd = frames.to(dev)
    H, W = frames.shape[-2:]
    ms = {}
    for L in sorted(set(lengths)):
        v = {"image": [f for f in fd[:L]], "target_hw": (H, W), "query_points": qp}
        one_step(model, v, L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step(model, v, L)
        torch.cuda.synchronize()
        ms[L] = (time.perf_counter() - t0) / steps * 1e3
    times = [ms[L] for L in lengths]
    total = sum(times)
    # least-squares line ms = a + b * frames: a is the fixed cost per clip the frame-count model of rounds 4 - 5 ignored
    n = len(ms)
    xs, ys = list(ms.keys()), list(ms.values())
    mx, my = sum(xs) / n, sum(ys) / n
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / max(sum((x - mx) ** 2 for x in xs), 1e-9)
    a = my - b * mx
    res = {"sequences": len(lengths), "frames_total": sum(lengths), "one_gpu_ms": round(total, 1),
           "one_gpu_fps": round(sum(lengths) / total * 1e3, 2),
           "measured_ms_by_length": {str(k): round(v, 1) for k, v in ms.items()},
           "fit_ms": {"fixed_per_clip": round(a, 1), "per_frame": round(b, 3)}, "by_world": {},
           "note": "model from per-sequence times measured on ONE GPU; LPT assignment by measured time"}
    for N in [int(v) for v in args.lpt_model.split(",") if v]:
        assign = lpt_assign([int(round(t * 1000)) for t in times], N)
        loads = [sum(times[i] for i in a_) for a_ in assign]
        by_frames = lpt_assign(lengths, N)
        loads_f = [sum(times[i] for i in a_) for a_ in by_frames]
        res["by_world"][str(N)] = {"predicted_ms": round(max(loads), 1), "predicted_speedup": round(total / max(loads), 2),
                                   "predicted_fps": round(sum(lengths) / max(loads) * 1e3, 1),
                                   "imbalance_max_over_mean": round(max(loads) / (total / N), 4),
                                   "predicted_speedup_lpt_by_frame_count": round(total / max(loads_f), 2)}
    return res


def self_launch(args):
    """``python bench.py --gpus N`` with N > 1 and no rendezvous in the environment: re-exec this very command line under
    ``torch.distributed.run`` (one rank per GPU, RCCL), exactly as the documented launch line does."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    from sam_pt_amd.dist import init_from_env
    from sam_pt_amd.synth import bench_clip
    rank, world, local = init_from_env()
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the hot path)"
    if world != args.gpus:
        assert world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch through torch.distributed.run"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # host-side weight generation / packing: keep N ranks from oversubscribing the host cores
    from sam_pt_amd.dist import host_threads
    torch.set_num_threads(host_threads(world))
    frames_sharded = args.shard == "frames" and world > 1
    lpt = args.shard == "lpt"
    lpt_info = None
    if lpt:            # every rank cuts its sequences out of one long synthetic clip (lengths: DAVIS-2017 val histogram)
        from sam_pt_amd.dist import DAVIS17_VAL_LENGTHS, lpt_assign
        lengths = DAVIS17_VAL_LENGTHS[:args.sequences]
        mine = [lengths[i] for i in lpt_assign(lengths, world)[rank]]
        loads = [sum(lengths[i] for i in a) for a in lpt_assign(lengths, world)]
        args.frames = max(lengths)
        lpt_info = {"sequences": len(lengths), "frames_total": sum(lengths), "frames_per_rank": loads,
                    "imbalance_max_over_mean": round(max(loads) / (sum(loads) / world), 4)}
    seed_a = 72 + (0 if (frames_sharded or lpt) else rank)
    frames, qp = bench_clip(T=args.frames, seed=seed_a, n_pos=args.points,
                            n_objects=args.objects, native=args.native_480p, n_neg=args.neg_points, square=args.square)
    # a SECOND clip (other textures; the scene geometry and hence the query points are the same) alternates with the first in
    # every timed loop: a step can reuse nothing of the previous one, and `clips_encoded` below proves each step ran its encoder
    frames_b, qp_b = bench_clip(T=args.frames, seed=seed_a + 1000, n_pos=args.points,
                                n_objects=args.objects, native=args.native_480p, n_neg=args.neg_points, square=args.square)
    assert torch.equal(qp, qp_b) and not torch.equal(frames, frames_b)
    H, W = frames.shape[-2:]
    model = build_model(args, dev)
    frames_dev, frames_b_dev = frames.to(dev), frames_b.to(dev)
    video = {"image": [f for f in frames_dev], "target_hw": (H, W), "query_points": qp}
    video_b = {"image": [f for f in frames_b_dev], "target_hw": (H, W), "query_points": qp_b}
    clips = [video, video_b]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    shard = "frames" if frames_sharded else "sequences"
    can_pipeline = not frames_sharded and not (lpt and world > 1)
    flight = ClipsInFlight(model, args.frames) if can_pipeline else None
    enc_stats = model.sam_predictor.stats

    def make_step(pipelined):
        def step(i):
            v0 = clips[i % 2]
            vs = [v0] if not lpt else [{**v0, "image": v0["image"][:L]} for L in mine]   # this rank's sequences
            m = None
            for v in vs:                                    # one SamPt.forward each (+ mask gather)
                m = flight.submit(v) if pipelined else one_step(model, v, args.frames, shard, args.shard_fnet)[0]
            return m
        return step

    if lpt and world > 1:                               # ranks hold different numbers of sequences: gather per step instead
        from sam_pt_amd.dist import gather_masks, index_masks

        def make_step(pipelined):                       # noqa: F811  (masks stay local; one gather of the last one per step)
            def step(i):
                m = None
                for L in mine:
                    out = model({**clips[i % 2], "image": clips[i % 2]["image"][:L]})
                    m = index_masks(torch.stack(out["logits"], dim=0))
                if m is None:
                    m = torch.zeros((0, H, W), dtype=torch.uint8, device=dev)
                gather_masks(m, args.frames)
                return m
            return step

    def timed_loop(pipelined):
        """W warm-up steps, then EXACTLY K steps between two barriers; max over ranks.  -> (seconds, last masks, frames encoded)"""
        step = make_step(pipelined)
        for i in range(args.warmup):
            step(i)
        if pipelined:
            flight.flush()                              # nothing of the warm-up is left in flight
        barrier()
        n0 = enc_stats["encoded_frames"]
        t0 = time.perf_counter()
        for i in range(args.steps):
            m = step(i)
        if pipelined:
            m = flight.flush()                          # the last clip is collected INSIDE the timed region
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, m, enc_stats["encoded_frames"] - n0

    value_pipelined_first = args.submit == "pipelined" and can_pipeline
    dt, masks, n_enc = timed_loop(value_pipelined_first)
    total_frames = (lpt_info["frames_total"] if lpt else (1 if frames_sharded else world) * args.frames) * args.steps
    fps = total_frames / dt
    # every timed step ran the image encoder over its own clip (nothing is keyed on a clip and reused): frames this rank encoded
    # inside the timed region == frames it was handed
    if lpt:
        frames_expected = sum(mine) * args.steps
    elif frames_sharded:
        from sam_pt_amd.dist import frame_batches
        frames_expected = sum(len(r) for r in frame_batches(args.frames, world, rank, max(1, min(8, -(-args.frames // world))))) * args.steps
    else:
        frames_expected = args.frames * args.steps
    assert n_enc == frames_expected, f"timed region encoded {n_enc} frames, expected {frames_expected}"
    clips_encoded = n_enc // max(args.frames, 1) if not (lpt or frames_sharded) else None
    # The same K steps once more the OTHER way (pipelined after blocking, or blocking after pipelined); same barriers, same max
    # over ranks, same alternating clips.
    fps_other = dt_other = None
    if can_pipeline and not args.no_pipelined:
        dt_other, _, n_enc2 = timed_loop(not value_pipelined_first)
        assert n_enc2 == frames_expected, (n_enc2, frames_expected)
        fps_other = total_frames / dt_other
    if value_pipelined_first:
        fps_pipe, dt_pipe, fps_seq, dt_seq = fps, dt, fps_other, dt_other
    else:
        fps_pipe, dt_pipe, fps_seq, dt_seq = fps_other, dt_other, fps, dt
    pipelined = value_pipelined_first
    insitu = None
    if rank == 0 and not args.no_roofline and args.precision in ("f16", "f16x3"):   # one more step, GEMM launches event-timed
        model.sam_predictor.gemm_profile_begin()
        one_step(model, video, args.frames) if world == 1 else model(video)
        torch.cuda.synchronize()
        insitu = model.sam_predictor.gemm_profile_end()
    if rank == 0:
        from sam_pt_amd.pack import fnet_f16x3_enabled
        tracker_precision = ("fp32-grade: mixer / update GEMMs exact f32 MFMA, correlation sampler fp32 VALU dot products (wave "
                             "shuffles, no MFMA); encoder convolutions 3-term split-fp16 MFMA (hi*hi + hi*lo + lo*hi, fp32 "
                             "accumulate)" if fnet_f16x3_enabled(args.tracker in ("pips", "cotracker"))
                             else "fp32 (GEMMs / convolutions exact f32 MFMA, correlation sampler fp32 VALU)")
        # BASELINE.json's metric string verbatim when the run IS that configuration; otherwise the same sentence with this run's
        # model / tracker / prompt so that a non-headline line cannot be mistaken for the headline
        headline = (args.model == "vit_h" and args.tracker == "pips" and args.points == 8 and args.objects == 1
                    and args.neg_points == 0 and not args.hq and not args.square)
        trk_name = {"pips": "PIPS", "cotracker": "CoTracker", "pips_plus_plus": "PIPS++"}[args.tracker]
        mdl_name = ("HQ-SAM " if args.hq else "") + {"vit_h": "ViT-H", "vit_l": "ViT-L", "vit_b": "ViT-B"}[args.model]
        metric = ("frames/sec end-to-end (ViT-H + PIPS, 480p, 8 pts, 1 obj) at 1/2/4/8 MI355X" if headline else
                  f"frames/sec end-to-end ({mdl_name} + {trk_name}, {'%d^2' % args.square if args.square else '480p'}, "
                  f"{args.points}{'+' + str(args.neg_points) if args.neg_points else ''} pts, {args.objects} obj) at 1/2/4/8 MI355X")
        res = {"metric": metric, "value": round(fps, 3),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong" if (frames_sharded or lpt) else "weak",
               "vs_baseline": None,
               # the metric as SURVEY.md §8(d) defines it — T / wall time of one blocking SamPt.forward — over K steps (this IS
               # `value` unless --submit pipelined), and the same K steps with one clip kept in flight
               "value_per_forward": round(fps_seq, 3) if fps_seq else None,
               "ms_per_forward": round(dt_seq / args.steps * 1e3, 2) if dt_seq else None,
               "value_pipelined": round(fps_pipe, 3) if fps_pipe else None,
               "ms_per_step_pipelined": round(dt_pipe / args.steps * 1e3, 2) if dt_pipe else None,
               "clips_encoded": clips_encoded, "distinct_clips_alternated": 2,
               "dtype": args.precision, "data": "synthetic",
               "config": {"workload": f"{'HQ-SAM' if args.hq else 'SAM'} {args.model} + { {'pips': 'PIPS', 'cotracker': 'CoTracker', 'pips_plus_plus': 'PIPS++'}[args.tracker]}, {args.points}"
                                      f"{'+' + str(args.neg_points) if args.neg_points else ''} query points, {args.objects} object(s), "
                                      + (f"{args.frames}x {H}x{W} synthetic frames, " if args.square else
                                         f"{args.frames}x 480p synthetic frames " + ("at native " if args.native_480p else "upscaled to ") + f"{H}x{W}, ") +
                                      f"{args.refine} refinement iterations, random-init weights (seed 72)",
                          "frames_per_step": lpt_info["frames_total"] if lpt else args.frames,
                          "submit": ("pipelined: one clip in flight ahead of the one being collected (SamPt.forward_begin / "
                                     "forward_end), all submitted and collected inside the timed region" if pipelined
                                     else "sequential: one blocking SamPt.forward per step"),
                          "parallelism": f"{'frame-batch' if frames_sharded else 'sequence'}-sharded x{world}",
                          "vit_precision": {"f16": "fp16 MFMA inputs, fp32 accumulate/LN/softmax/residual; patch embedding and neck "
                                                   "fp32-grade (3-term split-fp16 MFMA)",
                                            "f16x3": "fp32-grade on the fp16 matrix pipe: every GEMM / attention product as hi.hi + "
                                                     "hi.lo + lo.hi of split-fp16 operands, fp32 accumulate/LN/softmax/residual",
                                            "f32": "exact f32 MFMA, materialised attention scores"}[args.precision],
                          "tracker_precision": tracker_precision, "decoder_precision": "fp32"},
               "mask_foreground_fraction": round(float((masks > 0).float().mean()), 4),
               "published_reference_fps_unstated_hw": {"vit_h": 1.4, "vit_l": 1.8, "vit_b": 2.6}[args.model]}
        if lpt_info:
            res["lpt"] = lpt_info
        if world == 1 and not frames_sharded and not lpt:
            # where one blocking forward spends its wall time (GPU-side event times, ms after the start of the forward): tracker
            # encoder done, window chain done, image encoder done, decoder chain done (tools/forward_timeline.py)
            tl_ms = {}
            for _ in range(2):
                model.timeline = {}
                one_step(model, video, args.frames)
                torch.cuda.synchronize()
                e0 = model.timeline["start"][0]
                tl_ms = {k: round(e0.elapsed_time(ev), 1) for k, (ev, _) in model.timeline.items() if k != "start"}
            model.timeline = None
            res["timeline"] = {"unit": "ms after the forward's first launch (GPU events), one blocking forward", **tl_ms}
            lpr = getattr(model.point_tracker, "stats", {}).get("launches_per_round")
            if lpr:
                res["chain_launches_per_round"] = lpr
        if not args.no_roofline and args.precision in ("f16", "f16x3"):
            res["roofline"] = gemm_roofline(args, dev, insitu, (H, W))
            if args.precision == "f16":
                res["roofline"]["secondary"] = secondary_rooflines(args, dev)
        if world == 1 and not args.no_secondary:
            res["secondary"] = secondary_lines(args, model, video, dev)
            if headline and args.precision == "f16" and not args.no_roofline:
                # the reference-grade arithmetic (split-fp16: the reference's fp32 ViT at fp32 grade) as a FULL line of its own —
                # the same K timed steps, its own in-situ roofline block — from a child process running this very script
                import subprocess
                cmd = [sys.executable, os.path.abspath(__file__), "--precision", "f16x3", "--steps", str(args.steps), "--warmup",
                       str(args.warmup), "--no-cpu-baseline", "--no-secondary", "--no-pipelined"]
                try:
                    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                    line = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                    res["secondary_line"] = json.loads(line[-1]) if line else {"error": (pr.stderr or "no output")[-400:]}
                except Exception as ex:       # the headline line must not die with its side line
                    res["secondary_line"] = {"error": repr(ex)[:400]}
        if world == 1 and args.emulate_ranks:
            res["frame_sharding_model"] = frame_sharding_model(args, model, video)
        if world == 1 and args.lpt_model:
            res["sequence_sharding_model"] = sequence_sharding_model(args, model, dev)
        if world == 1 and not args.no_cpu_baseline:
            # the timed configuration's result, compared with the oracle: with pipelined submission the clip that had the
            # next one submitted on top of it (its decoder chain ran beside that one's tracker encoder)
            out = list(model.stream([video, video]))[0] if pipelined else model(video)
            torch.cuda.synchronize()
            res["cpu_baseline"], res["parity"] = cpu_reference(args, frames, qp, out)
        elif world == 1:
            out = list(model.stream([video, video]))[0] if pipelined else model(video)
            torch.cuda.synchronize()
            par = cached_parity(args, frames, qp, out)
            if par is not None:
                res["parity"] = par
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
