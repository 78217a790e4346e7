#!/usr/bin/env python
"""The per-clip decode chain alone (24 items x (1 + 12) passes, 576x1024, plain launches) — the workload for a kernel trace
of the decoder: rocprofv3 --kernel-trace -d <dir> -o dec -- python tools/decode_chain_trace.py [chains]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sam_pt_amd.sam_predictor import SamHip, SamPredictor

dev = torch.device("cuda:0")
# usage: decode_chain_trace.py [chains] [items F] [points K] [hq 0|1] [H] [W]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
F = int(sys.argv[2]) if len(sys.argv) > 2 else 24
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
hq = len(sys.argv) > 4 and sys.argv[4] == "1"
R, size = 12, ((int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (576, 1024))
pred = SamPredictor(SamHip("vit_b", seed=72, precision="f16", max_decode_batch=128, hq=hq).to(dev))   # decoder identical for B/L/H
pred._ensure()
st = pred.decode_staging(F, K, size)
g = torch.Generator().manual_seed(0)
st["feats"].copy_(torch.randn(F, 4096, 256, generator=g).to(dev) * 0.5)
st["pts"].copy_((torch.rand(F, K, 2, generator=g) * torch.tensor([1000.0, 560.0])).to(dev))
st["labels"].fill_(1)
f_in = st["feats"]
if hq:
    from sam_pt_amd.sam_predictor import ClipFeatures
    st["hq"].copy_(torch.randn(st["hq"].shape, generator=g).to(dev) * 0.1)
    f_in = ClipFeatures(st["feats"], st["hq"])
for _ in range(n):
    pred.track_decode(f_in, st["pts"], st["labels"], K, -1, R, -1e9, size, st["logits"], st["score"], graph=False)
torch.cuda.synchronize()
print("chains:", n)
