#!/usr/bin/env python
"""Which torch device ops (at::native kernels, small copies) one fused blocking forward still launches, and from which source line:
torch.profiler over one warm clip, every aten op that launched a device kernel or copy with its innermost sam_pt_amd frame.
python tools/stray_ops.py [bench.py options]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
from sam_pt_amd.synth import bench_clip  # noqa: E402

frames, qp = bench_clip(T=args.frames, seed=72, n_pos=args.points, n_objects=args.objects, n_neg=args.neg_points)
model = bench.build_model(args, dev)
fd = frames.to(dev)
video = {"image": [f for f in fd], "query_points": qp, "target_hw": tuple(frames.shape[-2:])}
for _ in range(3):
    model(video)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model(video)
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ops = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    if not ev.kernels:
        continue
    where = next((f for f in (ev.stack or []) if "sam_pt_amd" in f or "bench.py" in f), "?")
    ops[(ev.name, ", ".join(sorted({k.name[:50] for k in ev.kernels})), where.replace(root + "/", "")[:110])] += 1
print(f"{'calls':>5}  aten op -> device kernels  @ innermost frame of this repo")
for (name, kern, where), n in sorted(ops.items(), key=lambda kv: -kv[1]):
    print(f"{n:5d}  {name} -> {kern}  @ {where}")
print("total aten ops with device work:", sum(ops.values()))
