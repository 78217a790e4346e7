#!/usr/bin/env python
"""Where the REFERENCE protocol over the HIP seams (bench.py secondary.reference_sampt_over_hip_seams) spends its time:
cProfile of one SamPt.forward in call-by-call mode + GPU-side stage times.  python tools/ref_protocol_profile.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
from sam_pt_amd.sam_pt import SamPt  # noqa: E402
from sam_pt_amd.synth import bench_clip  # noqa: E402

frames, qp = bench_clip(T=args.frames, seed=72, n_pos=args.points, n_objects=args.objects)
model = bench.build_model(args, dev)
video = {"image": [f for f in frames.to(dev)], "target_hw": tuple(frames.shape[-2:]), "query_points": qp}
ref = SamPt(model.point_tracker, bench.ReferenceApiPredictor(model.sam_predictor), **bench.sampt_kwargs(args)).eval()
ref(video)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
ref(video)
torch.cuda.synchronize()
pr.disable()
print(f"forward: {time.perf_counter() - t0:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
