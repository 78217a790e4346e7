#!/usr/bin/env python
"""CPU emulation behind the fp16 ViT mode's static bias correction (tool, not product): how much of the
fp16 mode's embedding error goes away when the token-mean part of the weight-rounding error, mean_tokens(A).(W - fp16(W))^T, is
added back per GEMM — with the frame's own means, with the means of every 16th token, with means recorded on ANOTHER frame, and
with means calibrated on frames that share nothing with the test frame but the geometry (black, gray, uniform noise, another
synthetic scene, a square frame).  Built on tools/f16_error_budget.py (the oracle encoder with `.half().float()` at the fast
mode's rounding sites).   python tools/bias_correction_emulation.py [vit_b]

Result on the ViT-B bench frame (rms error of the embedding relative to the fp32 oracle): none 6.03e-4; own frame 3.94e-4; every
16th token 3.95e-4; another frame of the clip 3.95e-4; calibrated on: another scene 3.95e-4, uniform noise 4.0e-4, a square frame
5.2e-4, mid-gray 5.4e-4, black 6.9e-4.  The means follow the frame geometry and generic image statistics, so the product
calibrates once per geometry on a seeded noise frame (SamPredictor._select_bias_set)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch, torch.nn.functional as F
import f16_error_budget as E
from oracle import sam_ref as R
from sam_pt_amd.synth import bench_clip
from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
torch.set_num_threads(4)
MODE = "none"
STATIC = {}
orig_linear = F.linear
# the emulation hands F.linear an already rounded weight (r(W, site)): remember the original behind it
W_ORIG = {}
def r(x, site):
    if site in E.SITES:
        y = x.half().float()
        if site.startswith("w_"):
            W_ORIG[id(y)] = x
        return y
    return x
E.r = r
def linear(A, W, b=None):
    out = orig_linear(A, W, b)
    Wo = W_ORIG.pop(id(W), None)
    if Wo is None or MODE == "none":
        return out
    dW = Wo - W                                   # exact rounding residue
    shp = A.shape
    A2 = A.reshape(-1, shp[-1])
    if MODE == "frame":
        abar = A2.mean(0)
    elif MODE == "sub16":
        abar = A2[::16].mean(0)
    elif MODE in ("static_record", "static_apply"):
        key = (tuple(Wo.shape), float(Wo.flatten()[0]))
        if MODE == "static_record":
            STATIC[key] = A2.mean(0)
            abar = A2.mean(0)
        else:
            abar = STATIC[key]
    return out + orig_linear(abar[None], dW)[0]
E.F = type("Fx", (), {"linear": staticmethod(linear), "gelu": F.gelu, "conv2d": F.conv2d})
variant = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
cfg = SAM_CONFIGS[variant]
sd = init_sam_state_dict(cfg, 72)
frames, _ = bench_clip(T=6, seed=72)
sites = [s for s in E.ALL if s not in ("patch", "neck")]
with torch.no_grad():
    for fi in (0, 5):
        x = R.preprocess(cfg, frames[fi:fi+1].float())
        ref = R.image_encoder(sd, cfg, x)
        for mode in (["none", "frame", "sub16", "static_record"] if fi == 0 else ["none", "frame", "static_apply"]):
            MODE = mode
            E.SITES = set(sites)
            e = E.encoder(sd, cfg, x)
            d = (e - ref).double()
            print(f"frame {fi} {mode:14s} rel_max {float(d.abs().max()/ref.abs().max()):.3e} rel_rms {float(d.pow(2).mean().sqrt()/ref.double().pow(2).mean().sqrt()):.3e}", flush=True)

print("---- calibration robustness")
from sam_pt_amd.synth import synthetic_clip, upscale_to_longest_side
def calib_img(kind):
    if kind == "zeros": return torch.zeros(1, 3, 576, 1024, dtype=torch.uint8)
    if kind == "gray": return torch.full((1, 3, 576, 1024), 128, dtype=torch.uint8)
    if kind == "seed0_169":
        f, c = synthetic_clip(T=1, H=480, W=854, seed=0, disc_r=60.0); f, _ = upscale_to_longest_side(f, c, 1024); return f
    if kind == "seed0_square":
        f, c = synthetic_clip(T=1, H=1024, W=1024, seed=0, disc_r=72.0); return f
    if kind == "noise": 
        g = torch.Generator().manual_seed(0); return torch.randint(0, 256, (1, 3, 576, 1024), generator=g, dtype=torch.uint8)
with torch.no_grad():
    xb = R.preprocess(cfg, frames[5:6].float()); refb = R.image_encoder(sd, cfg, xb)
    for kind in ("zeros", "gray", "noise", "seed0_169", "seed0_square"):
        STATIC.clear(); MODE = "static_record"; E.SITES = set(sites)
        E.encoder(sd, cfg, R.preprocess(cfg, calib_img(kind).float()))
        MODE = "static_apply"
        e = E.encoder(sd, cfg, xb); d = (e - refb).double()
        print(f"calib {kind:14s} -> bench frame 5: rel_max {float(d.abs().max()/refb.abs().max()):.3e} rel_rms {float(d.pow(2).mean().sqrt()/refb.double().pow(2).mean().sqrt()):.3e}", flush=True)
