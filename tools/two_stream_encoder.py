#!/usr/bin/env python
"""Experiment: the SAM image encoder as TWO half-batch pipelines on two streams, each GEMM on half of the CUs.

The fp16 GEMM launches of one encoder batch reach their epilogues in lock step — all workgroups of a round write (and, for proj /
fc2, read) their 256 KB tiles at the same moment, at the chip's memory bandwidth, while the matrix cores wait (the
epilogue is 8 - 37 % of a launch) — and the LayerNorm / attention launches between them are memory- / latency-bound.  Two
independent pipelines over disjoint halves of the frames drift out of phase, so one pipeline's memory-bound stretches run beside
the other's MFMA-bound ones.  This tool measures exactly that, with no kernel change: one SamHip at batch B and W GEMM workgroups
per XCD against two SamHip instances (same weights) at batch B / 2 and W / 2 workgroups each on two streams.

  python tools/two_stream_encoder.py [model=vit_h] [frames=8] [wgs=32,28]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd.sam_predictor import SamHip, SamPredictor  # noqa: E402
from sam_pt_amd.synth import bench_clip  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "vit_h"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
WGS = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "32,28").split(",")]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
frames, _ = bench_clip(T=T, seed=72, n_pos=8, n_objects=1)
frames = frames.to(dev)
t0 = time.time()
from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict  # noqa: E402
cfg = SAM_CONFIGS[model]
sd = init_sam_state_dict(cfg, 72)
one = SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f16", max_batch=T).to(dev))
halves = [SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f16", max_batch=T // 2).to(dev)) for _ in range(2)]
print(f"models built in {time.time() - t0:.1f} s", flush=True)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
WARM, REPS = 3, 8


def timed(fn):
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


ref = one.encode_frames(frames)
ref = ref.emb if hasattr(ref, "emb") else ref
for w in WGS:
    def single(w=w):
        return one.encode_frames(frames, gemm_workgroups=w)

    def dual(w=w):
        cur = torch.cuda.current_stream()
        outs = []
        for s, p, fr in zip(streams, halves, (frames[:T // 2], frames[T // 2:])):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(p.encode_frames(fr, gemm_workgroups=w // 2))
        for s in streams:
            cur.wait_stream(s)
        return outs

    a = timed(single)
    b = timed(dual)
    o = dual()
    torch.cuda.synchronize()
    got = torch.cat([(x.emb if hasattr(x, "emb") else x) for x in o])
    same = bool(torch.equal(got, ref))
    print(f"{model} {T} frames, {w} GEMM workgroups per XCD: one pipeline {a:8.2f} ms | two half-batch pipelines ({w // 2} each) {b:8.2f} ms "
          f"({a / b:5.3f} x) | embeddings bitwise equal: {same}", flush=True)
