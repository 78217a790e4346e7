#!/usr/bin/env python
"""Per-clip decode chain (24 items x (1 + 12) passes, ViT-H geometry, 576x1024): plain launches vs hipGraph replay, timed
with HIP events on the chain's own (non-default) stream.  python tools/graph_vs_eager.py > profiles/<tag>_graph_vs_eager.log"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sam_pt_amd.sam_predictor import SamHip, SamPredictor
from sam_pt_amd.weights import SAM_CONFIGS

dev = torch.device("cuda:0")
F, K, R, size = 24, 8, 12, (576, 1024)
pred = SamPredictor(SamHip("vit_b", seed=72, precision="f16", max_decode_batch=128).to(dev))   # decoder identical for B/L/H
pred._ensure()
st = pred.decode_staging(F, K, size)
g = torch.Generator().manual_seed(0)
st["feats"].copy_(torch.randn(F, 4096, 256, generator=g).to(dev) * 0.5)
st["pts"].copy_((torch.rand(F, K, 2, generator=g) * torch.tensor([1000.0, 560.0])).to(dev))
st["labels"].fill_(1)
side = torch.cuda.Stream(device=dev)


def run(graph, n):
    with torch.cuda.stream(side):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            pred.track_decode(st["feats"], st["pts"], st["labels"], K, -1, R, -1e9, size, st["logits"], st["score"], graph=graph)
        side.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            pred.track_decode(st["feats"], st["pts"], st["labels"], K, -1, R, -1e9, size, st["logits"], st["score"], graph=graph)
        e1.record()
        t_host = (time.perf_counter() - t0) / n * 1e3
        side.synchronize()
        return e0.elapsed_time(e1) / n, t_host


for graph in (False, True, False, True):
    gpu_ms, host_ms = run(graph, 5)
    print(f"{'hipGraph replay' if graph else 'plain launches '}: {gpu_ms:7.2f} ms per chain on the GPU, {host_ms:6.2f} ms of host time to enqueue")
print("graph stats (cached, captures, replays):", pred.graph_stats())
