#!/usr/bin/env python
"""Where one SamPt.forward of the benchmark clip spends its wall time: GPU-side event times (ms after the start mark) and
the host clock at the same marks.  python tools/forward_timeline.py [bench.py options]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = bench.parse()
dev = torch.device("cuda:0")
from sam_pt_amd.synth import bench_clip
frames, qp = bench_clip(T=args.frames, seed=72, n_pos=args.points, n_objects=args.objects, native=args.native_480p,
                        n_neg=args.neg_points, square=args.square)
model = bench.build_model(args, dev)
video = {"image": [f for f in frames.to(dev)], "target_hw": tuple(frames.shape[-2:]), "query_points": qp}
for _ in range(3):
    model(video)
for rep in range(3):
    torch.cuda.synchronize()
    model.timeline = {}
    t0 = time.perf_counter()
    model(video)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tl = model.timeline
    e0, h0 = tl["start"]
    print(f"forward {1e3 * (t1 - t0):.1f} ms (start mark at +{1e3 * (h0 - t0):.1f} ms host) | " + " | ".join(
        f"{k}: gpu +{e0.elapsed_time(ev):.1f}, host +{1e3 * (h - h0):.1f}" for k, (ev, h) in tl.items() if k != "start"))
