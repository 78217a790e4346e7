#!/usr/bin/env python
"""The PIPS window rounds alone (feature pyramid prepared beforehand): wall time per clip, rounds, time per round.
python tools/tracker_bench.py [bench.py options]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
from sam_pt_amd.synth import bench_clip  # noqa: E402

frames, qp = bench_clip(T=args.frames, seed=72, n_pos=args.points, n_objects=args.objects, n_neg=args.neg_points)
model = bench.build_model(args, dev)
trk = model.point_tracker.to(dev)
rgbs = frames.to(dev)[None]
q = qp.reshape(1, -1, 3).to(dev)
for _ in range(2):
    trk.prepare(rgbs[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    trk.prepare(rgbs[0])
torch.cuda.synchronize()
print(f"tracker encoder (fnet + pyramid): {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per {args.frames}-frame clip")
for _ in range(2):
    trk(rgbs, q)
torch.cuda.synchronize()
w0 = trk.stats["windows"]
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    trk(rgbs, q)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps * 1e3
rounds = (trk.stats["windows"] - w0) / reps
print(f"tracker rounds alone: {dt:.1f} ms per {args.frames}-frame clip, {rounds:.0f} window rounds, {dt / max(rounds, 1):.2f} ms per round "
      f"({q.shape[1]} points)")
