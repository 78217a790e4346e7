#!/usr/bin/env python
"""The per-clip decode chain alone (24 items x (1 + 12) passes, 576x1024, plain launches) — the workload for a kernel trace
of the decoder: rocprofv3 --kernel-trace -d <dir> -o dec -- python tools/decode_chain_trace.py [chains]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sam_pt_amd.sam_predictor import SamHip, SamPredictor

dev = torch.device("cuda:0")
F, K, R, size = 24, 8, 12, (576, 1024)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pred = SamPredictor(SamHip("vit_b", seed=72, precision="f16", max_decode_batch=32).to(dev))   # decoder identical for B/L/H
pred._ensure()
st = pred.decode_staging(F, K, size)
g = torch.Generator().manual_seed(0)
st["feats"].copy_(torch.randn(F, 4096, 256, generator=g).to(dev) * 0.5)
st["pts"].copy_((torch.rand(F, K, 2, generator=g) * torch.tensor([1000.0, 560.0])).to(dev))
st["labels"].fill_(1)
for _ in range(n):
    pred.track_decode(st["feats"], st["pts"], st["labels"], K, -1, R, -1e9, size, st["logits"], st["score"], graph=False)
torch.cuda.synchronize()
print("chains:", n)
