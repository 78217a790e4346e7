#!/usr/bin/env python
"""CPU emulation of the fp16-ViT mode's rounding sites (tool, not product): which fp16 conversions of
csrc/engine_vit.hip cost how much of the embedding error against the fp32 oracle?

Every MFMA operand of the fast mode is fp16 (weights, LayerNorm outputs, q/k/v, softmax probabilities, attention output,
GELU output, the neck's inputs); accumulation, residual stream, LayerNorm and softmax are fp32.  The script re-runs the
oracle encoder (oracle/sam_ref.py) with `.half().float()` at the chosen sites and prints the relative error of the
(1,256,64,64) embedding per configuration.   python tools/f16_error_budget.py [vit_b] """
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import sam_ref as R
from sam_pt_amd.synth import bench_clip
from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict

SITES = set()


def r(x, site):
    return x.half().float() if site in SITES else x


def attention(sd, p, x, heads):
    B, H, W, D = x.shape
    hd = D // heads
    qkv = F.linear(r(x, "ln1"), r(sd[p + ".qkv.weight"], "w_qkv"), sd[p + ".qkv.bias"])
    qkv = r(qkv, "qkv").reshape(B, H * W, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    Rh = R._rel_table(r(sd[p + ".rel_pos_h"], "relpos"), H)
    Rw = R._rel_table(r(sd[p + ".rel_pos_w"], "relpos"), W)
    rq = q.reshape(B * heads, H, W, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(-1, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(-1, H * W, H * W)
    attn = attn - attn.max(-1, keepdim=True).values
    pe = r(attn.exp(), "p")                                   # flash kernel: P in fp16 for the PV MFMA, sum in fp32
    out = (pe @ v) / attn.exp().sum(-1, keepdim=True)
    out = out.view(B, heads, H, W, hd).permute(0, 2, 3, 1, 4).reshape(B, H, W, D)
    return F.linear(r(out, "att"), r(sd[p + ".proj.weight"], "w_proj"), sd[p + ".proj.bias"])


def block(sd, cfg, i, x):
    p = f"image_encoder.blocks.{i}"
    ws = 0 if i in cfg.global_attn_indexes else cfg.window_size
    sc = x
    x = R._ln(x, sd, p + ".norm1", 1e-6)
    if ws > 0:
        H, W = x.shape[1:3]
        x, pad = R._window_partition(x, ws)
    x = attention(sd, p + ".attn", x, cfg.num_heads)
    if ws > 0:
        x = R._window_unpartition(x, ws, pad, (H, W))
    x = sc + x
    y = R._ln(x, sd, p + ".norm2", 1e-6)
    y = F.gelu(F.linear(r(y, "ln2"), r(sd[p + ".mlp.lin1.weight"], "w_fc1"), sd[p + ".mlp.lin1.bias"]))
    y = F.linear(r(y, "hid"), r(sd[p + ".mlp.lin2.weight"], "w_fc2"), sd[p + ".mlp.lin2.bias"])
    return x + y


def encoder(sd, cfg, x):
    w = sd["image_encoder.patch_embed.proj.weight"]
    x = F.conv2d(r(x, "patch"), r(w, "patch"), sd["image_encoder.patch_embed.proj.bias"], stride=cfg.patch_size).permute(0, 2, 3, 1)
    x = x + sd["image_encoder.pos_embed"]
    for i in range(cfg.depth):
        x = block(sd, cfg, i, x)
    x = x.permute(0, 3, 1, 2)
    x = R._ln2d(F.conv2d(r(x, "neck"), r(sd["image_encoder.neck.0.weight"], "neck")), sd, "image_encoder.neck.1")
    x = R._ln2d(F.conv2d(r(x, "neck"), r(sd["image_encoder.neck.2.weight"], "neck"), padding=1), sd, "image_encoder.neck.3")
    return x


ALL = ["ln1", "w_qkv", "qkv", "relpos", "p", "att", "w_proj", "ln2", "w_fc1", "hid", "w_fc2", "patch", "neck"]


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
    cfg = SAM_CONFIGS[variant]
    sd = init_sam_state_dict(cfg, 72)
    frames, _ = bench_clip(T=1, seed=72)
    x = R.preprocess(cfg, frames.float())
    with torch.no_grad():
        ref = R.image_encoder(sd, cfg, x)
        def run(sites, label):
            global SITES
            SITES = set(sites)
            t0 = time.time()
            e = encoder(sd, cfg, x)
            d = (e - ref).double()
            print(f"{label:40s} rel_max {float(d.abs().max() / ref.abs().max()):.3e}  rel_rms {float(d.pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()):.3e}  ({time.time() - t0:.0f}s)", flush=True)
        run([], "none (sanity)")
        run(ALL, "all sites (the fast mode)")
        for s in ALL:
            run([s], "only " + s)
        run([s for s in ALL if s != "neck"], "all but neck")
        run([s for s in ALL if not s.startswith("w_")], "activations only (weights exact)")
        run([s for s in ALL if s.startswith("w_")], "weights only")


if __name__ == "__main__":
    main()
