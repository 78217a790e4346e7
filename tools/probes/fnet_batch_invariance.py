#!/usr/bin/env python
"""Is the tracker encoder's output for a frame independent of the batch it is encoded in?  Pyramids of one clip with encoder chunks of
4 / 3 / 1 frames, compared bit for bit (the frame-sharded mode relies on it).   python tools/probes/fnet_batch_invariance.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from sam_pt_amd.point_tracker import CoTrackerPointTracker  # noqa: E402
from tests.util import synthetic_clip  # noqa: E402

dev = torch.device("cuda:0")
frames, _ = synthetic_clip(T=10, H=128, W=256, seed=4)
fr = frames.to(dev)
ref = None
for chunk in (4, 4, 3, 1):
    trk = CoTrackerPointTracker(seed=72, fnet_chunk=chunk).to(dev)
    pyr = trk.compute_pyramid(fr)
    torch.cuda.synchronize()
    if ref is None:
        ref = [p.clone() for p in pyr]
        continue
    for l, (a, b) in enumerate(zip(pyr, ref)):
        d = (a - b).abs().amax(dim=(1, 2, 3))
        print(f"chunk {chunk} vs 4, level {l}: max |diff| per frame", [f"{v:.1e}" for v in d.tolist()])
