"""Host-side weight repacking: upstream-layout state dicts -> the named device buffers the HIP engines expect.

This is load-time plumbing (runs once per model on the CPU, then uploads): no hot-path arithmetic happens here.

Layouts (see sam_pt_amd/csrc/engine_*.hip):
  * conv weights   torch [Cout][Cin][KH][KW]  ->  [Cout][KH][KW][Cin]  (implicit-GEMM K order, NHWC activations)
  * PIPS stem      Cin zero-padded 3 -> 4 so that one float4 is one pixel
  * mixer input    Linear(519 -> 512) weight K-padded to 520 (16-byte rows)
  * ViT GEMMs      fp16 copies under "<key>.f16" for the fast mode, fp32 originals for the exact mode
  * ConvTranspose  torch [Cin][Cout][2][2] -> [(dy,dx)][Cout][Cin] + pixel-shuffle row maps
"""
from __future__ import annotations

import os
from typing import Dict

import torch

from .weights import SamConfig


def _khwc(w: torch.Tensor, pad_cin_to: int = 0) -> torch.Tensor:
    w = w.permute(0, 2, 3, 1).contiguous()  # [Cout][KH][KW][Cin]
    if pad_cin_to and w.shape[-1] < pad_cin_to:
        w = torch.nn.functional.pad(w, (0, pad_cin_to - w.shape[-1]))
    return w.reshape(w.shape[0], -1).contiguous()


F16X3_WSHIFT = 8          # csrc/common.h


def split_f16x3(w: torch.Tensor, strict: bool = True):
    """fp32 weights [Cout][K] -> half [2][Cout][K]: w * 2^8 = hi + lo with hi = fp16(.), lo = fp16(. - hi), the operand
    format of csrc/conv_f16x3.hip (three fp16 MFMAs reproduce the fp32 product to 2^-22).  A weight with |w| * 2^8 outside
    the fp16 range cannot be split: ``strict`` raises, otherwise None is returned and the caller leaves the planes out
    (every decoder / tracker-encoder consumer then runs that one layer on the exact f32 MFMA path)."""
    ws = w.float() * float(1 << F16X3_WSHIFT)
    if not bool(torch.isfinite(ws).all()) or float(ws.abs().max()) >= 65504.0:
        if strict:
            raise ValueError(f"split_f16x3: weight magnitude {float(w.abs().max()):.3g} * 2^{F16X3_WSHIFT} outside the fp16 "
                             "range (this layer has no f32 fallback in the fp16-ViT mode: use precision='f32')")
        return None
    hi = ws.half()
    lo = (ws - hi.float()).half()
    return torch.stack([hi, lo]).contiguous()


def x3_rows(x: torch.Tensor, scale_shift: int = 0) -> torch.Tensor:
    """fp32 rows [R][K] (K % 32 == 0) -> half [R][2K] "x3 rows", the operand format of the split-fp16 GEMM / attention
    kernels (csrc/common.h GemmP::x3): every block of 32 consecutive k is stored as hi(32) | lo(32) with
    v * 2^scale_shift = hi + lo, hi = fp16(.) clamped to the fp16 range, lo = fp16(. - hi).  Weights use
    ``scale_shift = F16X3_WSHIFT`` (so that lo stays in fp16's normal range; the GEMM multiplies by 2^-8), activations 0."""
    R, K = x.shape
    assert K % 32 == 0, f"x3_rows: K = {K} is not a multiple of 32"
    v = x.float() * float(1 << scale_shift)
    hi = v.clamp(-65504.0, 65504.0).half()
    lo = (v - hi.float()).half()
    if not bool(torch.isfinite(lo.float()).all()):
        raise ValueError(f"x3_rows: magnitude {float(x.abs().max()):.3g} * 2^{scale_shift} cannot be split into two fp16 pieces")
    return torch.stack([hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)], dim=2).reshape(R, 2 * K).contiguous()


def x3_unrows(y: torch.Tensor, scale_shift: int = 0) -> torch.Tensor:
    """Inverse of ``x3_rows``: half [R][2K] -> fp32 [R][K] = (hi + lo) * 2^-scale_shift (tests / debugging)."""
    R, K2 = y.shape
    p = y.float().view(R, K2 // 64, 2, 32)
    return ((p[:, :, 0] + p[:, :, 1]).reshape(R, K2 // 2) / float(1 << scale_shift)).contiguous()


def _put_split(out: Dict[str, torch.Tensor], key: str, w: torch.Tensor) -> None:
    """out[key] = split planes of w — or nothing when w cannot be split (the consumer falls back to f32 for that layer)."""
    hl = split_f16x3(w, strict=False)
    if hl is not None:
        out[key] = hl


def fnet_f16x3_enabled(default: bool) -> bool:
    """Whether the tracker encoder's convolutions run as 3-term split-fp16 MFMAs (csrc/conv_f16x3.hip) or as exact fp32
    MFMAs.  ``SAMPT_FNET_F16X3=0|1`` overrides the per-tracker default (PIPS: on, PIPS++: off — see DESIGN.md)."""
    v = os.environ.get("SAMPT_FNET_F16X3")
    return default if v is None or v == "" else v != "0"


def _add_fnet_split(out: Dict[str, torch.Tensor], sd: Dict[str, torch.Tensor], default: bool) -> None:
    """``<conv>.weight_hl`` for every encoder convolution whose Cin is a multiple of 32 (all but the 3-channel stem)."""
    if not fnet_f16x3_enabled(default):
        return
    for k, v in sd.items():
        if k.startswith("fnet.") and k.endswith(".weight") and v.dim() == 4 and v.shape[1] % 32 == 0:
            _put_split(out, k + "_hl", out[k])


MIXER_X3_ASHIFT = 6       # csrc/pips_mixer_x3.hip: activations are split as v * 2^6 = hi + lo (lo stays a normal fp16 down to |v| = 2e-3)
MIXER_X3_SLICES = (16, 32)   # hidden slices the split-fp16 mixer kernels exist for (8 / 4 fragments of 16 hidden units per slice)


def pips_mixer_x3_stream(w1: torch.Tensor, w2: torch.Tensor, NS: int):
    """Channel-MLP weights of one mixer block (w1 [2048][512], w2 [512][2048], fp32) -> half [NS][NF * 64 * 512], NF = 2048 / NS / 16:
    per hidden slice the LINEAR stream of 1-KB MFMA operand images k_pips_mix_mlp_x3 consumes, in consumption order, so that every
    LDS-DMA instruction of the kernel copies 1 KB of contiguous memory and every wave reads its operand back at lane * 16.
    An image is [lane 0..63][8 halves]: lane (lr = lane & 15, lq = lane >> 4) of v_mfma_f32_16x16x32_f16 supplies row lr, k slots
    8 lq .. 8 lq + 7.  Weights are scaled by 2^8 and split into hi / lo fp16 planes (``split_f16x3``).
      fc1: for ks (16 steps of 32 k), fragment f, plane:  image[lane][e] = W1[h0 + 16 f + lr][32 ks + 8 lq + e]
      fc2: for output fragment o (32), k pair kp (NF / 2), plane:  image[lane][e] = W2[16 o + lr][h0 + 32 kp + perm(lq, e)] with
           perm(lq, e) = 4 lq + e (e < 4) | 16 + 4 lq + e - 4 (e >= 4): the k slots of the hidden activations as the first product's
           accumulators hold them (fragments 2 kp and 2 kp + 1, four consecutive hidden units per lane each).
    Returns None when a weight is outside the splittable range (the engine then keeps the exact-f32 kernels)."""
    H, D = w1.shape
    assert w2.shape == (D, H) and H % (16 * NS) == 0 and D % 32 == 0
    p1, p2 = split_f16x3(w1, strict=False), split_f16x3(w2, strict=False)
    if p1 is None or p2 is None:
        return None
    NF = H // NS // 16
    # pure permutations of the planes (lane = 16 lq + lr):
    #   fc1  [plane][NS][f][lr][ks][lq][e]            -> [NS][ks][f][plane][lq][lr][e]
    #   fc2  [plane][o][lr][NS][kp][half][lq][e4]     -> [NS][o][kp][plane][lq][lr][half][e4]      (e = 4 half + e4)
    img1 = p1.view(2, NS, NF, 16, D // 32, 4, 8).permute(1, 4, 2, 0, 5, 3, 6).reshape(NS, -1)
    img2 = p2.view(2, D // 16, 16, NS, NF // 2, 2, 4, 4).permute(3, 1, 4, 0, 6, 2, 5, 7).reshape(NS, -1)
    return torch.cat([img1, img2], dim=1).contiguous()


stats = {"packs": 0}     # weight-packing calls of this process (tests: none may happen inside a timed step loop)


def pack_pips(sd: Dict[str, torch.Tensor], device, S: int = 8) -> Dict[str, torch.Tensor]:
    stats["packs"] += 1
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        v = v.detach().float()
        if k.startswith("fnet.") and k.endswith(".weight"):
            out[k] = _khwc(v, pad_cin_to=4 if k == "fnet.conv1.weight" else 0)
        elif k == "delta_block.to_delta.0.weight":
            out[k] = torch.nn.functional.pad(v, (0, 520 - v.shape[1])).contiguous()
        elif ".0.fn." in k and k.endswith(".weight"):
            out[k] = v.reshape(v.shape[0], v.shape[1]).contiguous()  # Conv1d(k=1) -> [out][in]
        else:
            out[k] = v.contiguous()
    out["ffeat_updater.0.weight_t"] = sd["ffeat_updater.0.weight"].detach().float().t().contiguous()
    out["vis_predictor.0.weight"] = sd["vis_predictor.0.weight"].detach().float().reshape(-1).contiguous()
    out["__times"] = torch.linspace(0, S, S)  # pips.py:527
    _add_fnet_split(out, sd, default=True)
    # channel-MLP weights of the 12 mixer blocks as split-fp16 operand streams (csrc/pips_mixer_x3.hip), one per supported slice
    # count; left out when a weight cannot be split (the engine then keeps the exact-f32 kernels) or with SAMPT_PIPS_MIXER_X3=0
    if os.environ.get("SAMPT_PIPS_MIXER_X3", "1") != "0":
        streams = {}
        for i in range(1, 13):
            p = f"delta_block.to_delta.{i}"
            if p + ".1.fn.0.weight" not in sd:
                streams = None
                break
            for ns in MIXER_X3_SLICES:
                st = pips_mixer_x3_stream(out[p + ".1.fn.0.weight"], out[p + ".1.fn.3.weight"], ns) if streams is not None else None
                if st is None:
                    streams = None
                    break
                streams[f"{p}.__x3s{ns}"] = st
            if streams is None:
                break
        if streams:
            out.update(streams)
    return {k: v.to(device) for k, v in out.items()}


def sincos_1d(embed_dim: int, pos) -> torch.Tensor:
    """CoTracker's ``get_1d_sincos_pos_embed_from_grid`` (the MAE recipe): (M,) positions -> (M, embed_dim) =
    [sin(pos * omega), cos(pos * omega)], omega_d = 10000^(-d / (embed_dim / 2)), evaluated in float64 and cast to f32 as
    upstream does (numpy float64 -> ``.float()``)."""
    import numpy as np
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", np.asarray(pos, dtype=np.float64).reshape(-1), omega)
    return torch.from_numpy(np.concatenate([np.sin(out), np.cos(out)], axis=1)).float()


def cotracker_pos_tables(H0: int, W0: int, embed_dim: int = 456):
    """The two 1-D tables of ``get_2d_sincos_pos_embed(embed_dim, (H0, W0))``: channels [0, E/2) of grid cell (y, x) are
    ``pos_x[x]``, channels [E/2, E) are ``pos_y[y]`` (upstream's meshgrid puts the column grid first)."""
    return sincos_1d(embed_dim // 2, range(W0)).contiguous(), sincos_1d(embed_dim // 2, range(H0)).contiguous()


def pack_cotracker(sd: Dict[str, torch.Tensor], device, S: int = 8) -> Dict[str, torch.Tensor]:
    """CoTracker v1: encoder convolutions as for PIPS; the UpdateFormer's Linear weights are used as
    they are ([out][in] rows); ``__times_embed`` = the 1-D sin/cos embedding of the S window frames (456 channels);
    ``__ln_ones`` / ``__ln_zeros``: the blocks' LayerNorms are affine-free."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        v = v.detach().float()
        if k.startswith("fnet.") and k.endswith(".weight"):
            out[k] = _khwc(v, pad_cin_to=4 if k == "fnet.conv1.weight" else 0)
        else:
            out[k] = v.contiguous()
    out["ffeat_updater.0.weight_t"] = sd["ffeat_updater.0.weight"].detach().float().t().contiguous()
    out["vis_predictor.0.weight"] = sd["vis_predictor.0.weight"].detach().float().reshape(-1).contiguous()
    hidden = sd["updateformer.input_transform.weight"].shape[0]
    out["__times_embed"] = sincos_1d(sd["updateformer.input_transform.weight"].shape[1], torch.linspace(0, S - 1, S).numpy())
    out["__ln_ones"], out["__ln_zeros"] = torch.ones(hidden), torch.zeros(hidden)
    _add_fnet_split(out, sd, default=True)
    return {k: v.to(device) for k, v in out.items()}


def pack_pips2(sd: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """PIPS++ (pips_plus_plus.py:420-434): fnet convs as for PIPS; DeltaBlock Conv1d weights (Cout, Cin, 3) ->
    [Cout][3][Cin] (tap-major, channel fastest = the implicit-GEMM K order over an [n][S][1][C] image), the first conv's
    718 input channels zero-padded to 720; ``__omega`` = the 32 frequencies of posemb_sincos_2d_xy (misc.py:18-19)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        v = v.detach().float()
        if k.startswith("fnet.") and k.endswith(".weight"):
            out[k] = _khwc(v, pad_cin_to=4 if k == "fnet.conv1.weight" else 0)
        elif k.startswith("delta_block.") and k.endswith(".conv.weight"):
            w = v.permute(0, 2, 1).contiguous()                                   # (Cout, 3, Cin)
            if k == "delta_block.first_block_conv.conv.weight":
                w = torch.nn.functional.pad(w, (0, 720 - w.shape[-1]))
            out[k] = w.reshape(w.shape[0], -1).contiguous()
        else:
            out[k] = v.contiguous()
    omega = torch.arange(32) / 31
    out["__omega"] = (1.0 / (10000 ** omega)).float()
    _add_fnet_split(out, sd, default=False)
    return {k: v.to(device) for k, v in out.items()}


def window_row_map(grid: int, window: int, batches: int, rows: int = 0) -> torch.Tensor:
    """Row map of SAM's window_partition (App. A-3): entry ((b*nwin + w)*window^2 + i) = source token row
    b*grid^2 + y*grid + x, or -1 where the window hangs over the zero padding.  `rows` > 0 (a multiple of `window`):
    the map of a token stream that holds only the first `rows` token rows of every frame (rows x grid tokens per frame,
    rows/window x ceil(grid/window) windows) — the compact stream of the encoder's dead-row skipping."""
    n1 = (grid + window - 1) // window
    gh = rows if rows > 0 else grid
    ny = (gh + window - 1) // window
    ys = torch.arange(n1 * window).view(n1, window)
    m = torch.full((ny, n1, window, window), -1, dtype=torch.int64)
    for wy in range(ny):
        for wx in range(n1):
            yy = ys[wy].view(window, 1).expand(window, window)
            xx = ys[wx].view(1, window).expand(window, window)
            ok = (yy < gh) & (xx < grid)
            m[wy, wx] = torch.where(ok, yy * grid + xx, torch.full_like(yy, -1))
    m = m.reshape(1, -1).repeat(batches, 1)
    off = (torch.arange(batches) * gh * grid).view(batches, 1)
    m = torch.where(m >= 0, m + off, m)
    return m.reshape(-1).to(torch.int32)


def rel_pos_operand_images(rel_h: torch.Tensor, rel_w: torch.Tensor) -> torch.Tensor:
    """The decomposed rel-pos tables of one block ([2 S - 1][hd] each) as the fp16 MFMA operand images the flash kernel's prologue
    multiplies with the query fragments (csrc/attention.hip: G[rho][q] = <rel_pos[rho], q>): half [2 tables][tiles of 32 rows][hd / 16
    k-steps][64 lanes][8], lane l of tile t, k-step ks holds rel_pos[32 t + (l & 31)][16 ks + 8 (l >> 5) .. + 8] (rows past the table:
    0).  The kernel used to build them from the f32 tables in every workgroup: 11 % of a windowed launch (profiles/r6_c43_*)."""
    assert rel_h.shape == rel_w.shape and rel_h.shape[1] % 16 == 0
    nr, hd = rel_h.shape
    nt, ks = -(-nr // 32), hd // 16
    lane = torch.arange(64)
    row = torch.arange(nt)[:, None, None] * 32 + (lane & 31)[None, None, :]                                   # [nt][1][64]
    col = torch.arange(ks)[None, :, None, None] * 16 + (lane >> 5)[None, None, :, None] * 8 + torch.arange(8)  # [1][ks][64][8]
    out = torch.zeros(2, nt, ks, 64, 8, dtype=torch.float16)
    for i, tab in enumerate((rel_h, rel_w)):
        t = torch.cat([tab.float(), torch.zeros(nt * 32 - nr, hd)])
        out[i] = t[row[..., None].expand(nt, ks, 64, 8), col.expand(nt, ks, 64, 8)].half()
    return out.contiguous()


def pack_vit(sd: Dict[str, torch.Tensor], cfg: SamConfig, device, f16, win_batches: int) -> Dict[str, torch.Tensor]:
    """``f16``: False / 0 = exact fp32, True / 1 = fp16 block GEMMs (".f16" copies), 2 = split-fp16 block GEMMs (".x3" x3 rows
    of w * 2^8); the encoder's two ends are split-fp16 planes ("_hl") in both 16-bit modes."""
    stats["packs"] += 1
    out: Dict[str, torch.Tensor] = {}
    mode = int(f16)
    f16 = mode != 0
    e = "image_encoder."
    # The two ends of the encoder stay fp32-grade in the fast mode too (3-term split-fp16 MFMAs, 0.3 % of the FLOPs): every fp16
    # rounding inside the 12-32 blocks is damped by the residual stream, but the neck's roundings land on the embedding
    # directly — they were the largest single term (30 % of the error variance) of the fp16 mode's error budget
    # (tools/f16_error_budget.py), the patch embedding another 7 %.
    exact_keys = [e + "patch_embed.proj.weight", e + "neck.0.weight"]
    gemm_keys = []
    for i in range(cfg.depth):
        p = f"{e}blocks.{i}."
        gemm_keys += [p + "attn.qkv.weight", p + "attn.proj.weight", p + "mlp.lin1.weight", p + "mlp.lin2.weight"]
    for k, v in sd.items():
        if not k.startswith(e):
            continue
        v = v.detach().float()
        if k == e + "pos_embed":
            out[k] = v.reshape(-1, v.shape[-1]).contiguous()
        elif k == e + "neck.2.weight":
            w = _khwc(v)
            if f16:      # fast mode: fp32-grade on the fp16 matrix pipe (3-term split, csrc/conv_f16x3.hip)
                out[e + "neck.2.weight_khwc_hl"] = split_f16x3(w)
            else:
                out[e + "neck.2.weight_khwc"] = w
        elif k in exact_keys:
            w = v.reshape(v.shape[0], -1).contiguous()
            if f16:
                out[k + "_hl"] = split_f16x3(w)
            else:
                out[k] = w
        elif f16 and k.endswith(".attn.qkv.bias"):
            # the 16-bit attention kernels read the qkv of SAM's zero-padded window tokens (= the bias) from ONE row in the qkv
            # matrix's own format instead of from filled-in rows (csrc/ops.h FlashPad)
            out[k] = v.contiguous()
            out[k + (".x3" if mode == 2 else ".f16")] = x3_rows(v.view(1, -1)).view(-1) if mode == 2 else v.half().contiguous()
        elif k in gemm_keys:
            w = v.reshape(v.shape[0], -1).contiguous()
            if mode == 2:
                out[k + ".x3"] = x3_rows(w, F16X3_WSHIFT)
            else:
                out[k + (".f16" if f16 else "")] = w.half() if f16 else w
        else:
            out[k] = v.contiguous()
    if mode == 1 and os.environ.get("SAMPT_ATTN_REL_OPS", "1") != "0":    # fp16 attention: the rel-pos tables as ready-made operand images
        for i in range(cfg.depth):
            p = f"{e}blocks.{i}.attn."
            if p + "rel_pos_h" in out and out[p + "rel_pos_h"].shape[1] % 16 == 0:
                out[p + "rel_pos_ops"] = rel_pos_operand_images(out[p + "rel_pos_h"], out[p + "rel_pos_w"])
    rows = window_row_map(cfg.grid, cfg.window_size, win_batches)
    out["__win_rows"] = rows
    # token row -> row in window order (a permutation into the padded layout), and the padded rows themselves: the
    # qkv / proj GEMMs of windowed blocks then run on the REAL tokens only (padding adds 20 % rows at grid 64, window 14)
    valid = (rows >= 0).nonzero().flatten()
    inv = torch.empty(win_batches * cfg.grid * cfg.grid, dtype=torch.int32)
    inv[rows[valid].long()] = valid.to(torch.int32)
    out["__win_inv"] = inv
    out["__win_pad"] = (rows < 0).nonzero().flatten().to(torch.int32)
    # the same two maps for the compact streams of frames that fill only k window rows of the padded square (16:9 video:
    # k = 3 of 5) — VitEngine::encode runs the blocks before the first global one on those rows only
    for k in range(1, min(8, cfg.grid // cfg.window_size) + 1):
        lh = k * cfg.window_size
        if lh >= cfg.grid:
            break
        r = window_row_map(cfg.grid, cfg.window_size, win_batches, rows=lh)
        v = (r >= 0).nonzero().flatten()
        iv = torch.empty(win_batches * lh * cfg.grid, dtype=torch.int32)
        iv[r[v].long()] = v.to(torch.int32)
        out[f"__win_inv_live{k}"] = iv
        out[f"__win_pad_live{k}"] = (r < 0).nonzero().flatten().to(torch.int32)
    return {k: v.to(device) for k, v in out.items()}


def dense_pe(gauss: torch.Tensor, grid: int) -> torch.Tensor:
    """PromptEncoder.get_dense_pe() (App. A-4) as a token-major [grid*grid][256] constant."""
    import math
    ones = torch.ones(grid, grid)
    y = (ones.cumsum(0) - 0.5) / grid
    x = (ones.cumsum(1) - 0.5) / grid
    c = 2 * torch.stack([x, y], dim=-1) - 1
    c = 2 * math.pi * (c @ gauss)
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).reshape(grid * grid, -1).contiguous()


def _convt_pack(w: torch.Tensor) -> torch.Tensor:
    # torch ConvTranspose2d weight [Cin][Cout][2][2] -> [(dy,dx)][Cout][Cin]
    return w.permute(2, 3, 1, 0).reshape(4, w.shape[1], w.shape[0]).contiguous()


def _shuffle_map(side: int, frames: int = 1) -> torch.Tensor:
    """Pixel-shuffle row maps of a stride-2 ConvTranspose2d for a batch of `frames` maps: [4][frames*side*side];
    input pixel p = y*side + x of frame f -> output row f*4*side^2 + (2y+dy)*(2*side) + 2x+dx, (dy,dx) in order."""
    y = torch.arange(side).view(side, 1).expand(side, side)
    x = torch.arange(side).view(1, side).expand(side, side)
    off = (torch.arange(frames) * 4 * side * side).view(frames, 1)
    maps = [(((2 * y + dy) * (2 * side) + 2 * x + dx).reshape(1, -1) + off).reshape(-1) for dy in range(2) for dx in range(2)]
    return torch.stack(maps).to(torch.int32).contiguous()


def pack_decoder(sd: Dict[str, torch.Tensor], cfg: SamConfig, device, max_frames: int = 1,
                 hq: bool = False) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.startswith("mask_decoder.") or k.startswith("prompt_encoder."):
            out[k] = v.detach().float().contiguous()
    toks = [sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]]
    if hq:
        toks.append(sd["mask_decoder.hf_token.weight"])                       # 6th output token (MaskDecoderHQ)
        for p in ("compress_vit_feat", "embedding_encoder"):
            for i in (0, 3):
                out[f"mask_decoder.{p}.{i}.weight_packed"] = _convt_pack(sd[f"mask_decoder.{p}.{i}.weight"].float())
        for i in (0, 3):                                                       # conv3x3 OIHW -> [O][(ky,kx,ci)]
            w = sd[f"mask_decoder.embedding_maskfeature.{i}.weight"].float()
            out[f"mask_decoder.embedding_maskfeature.{i}.weight_packed"] = \
                w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
            if os.environ.get("SAMPT_DEC_F16X3", "1") != "0":                  # fp32-grade on the fp16 pipe (Cin 32 / 64)
                _put_split(out, f"mask_decoder.embedding_maskfeature.{i}.weight_packed_hl",
                           out[f"mask_decoder.embedding_maskfeature.{i}.weight_packed"])
    out["mask_decoder.__out_tokens"] = torch.cat(toks, dim=0).float().contiguous()
    out["prompt_encoder.__point_embeddings"] = torch.cat(
        [sd[f"prompt_encoder.point_embeddings.{i}.weight"] for i in range(4)], dim=0).float().contiguous()
    g = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].float()
    out["prompt_encoder.__dense_pe"] = dense_pe(g, cfg.grid)
    out["mask_decoder.output_upscaling.0.weight_packed"] = _convt_pack(sd["mask_decoder.output_upscaling.0.weight"].float())
    out["mask_decoder.output_upscaling.3.weight_packed"] = _convt_pack(sd["mask_decoder.output_upscaling.3.weight"].float())
    if os.environ.get("SAMPT_DEC_F16X3", "1") != "0":
        # the transposed convolutions as ONE split-fp16 GEMM per stage over all four sub-pixels: planes [2][4*cout][cin]
        names = ["output_upscaling"] + (["compress_vit_feat", "embedding_encoder"] if hq else [])
        for nme in names:
            for i in (0, 3):
                wp = out[f"mask_decoder.{nme}.{i}.weight_packed"]
                if wp.shape[2] % 32 == 0 and wp.shape[1] % 4 == 0:
                    _put_split(out, f"mask_decoder.{nme}.{i}.weight_packed_hl", wp.reshape(-1, wp.shape[2]))
    # Fused image-side projections of the two-way transformer (csrc/engine.h DecEngine::FusedProj): the token -> image keys
    # (keys + pe) Wk, values keys Wv and the image -> token queries (keys + pe) Wq' of a layer read the same image tokens, so
    # they run as one GEMM with W = [Wk; Wv; Wq'] and the constant pe [Wk; 0; Wq']^T added in its epilogue.
    tr = "mask_decoder.transformer."
    pe64 = out["prompt_encoder.__dense_pe"].double()

    def fuse(key, mats, with_pe):
        W = torch.cat([out[m + ".weight"] for m in mats], dim=0)
        out[key + "_w"] = W.contiguous()
        out[key + "_b"] = torch.cat([out[m + ".bias"] for m in mats], dim=0).contiguous()
        Wpe = torch.cat([out[m + ".weight"] if u else torch.zeros_like(out[m + ".weight"]) for m, u in zip(mats, with_pe)], dim=0)
        out[key + "_pe"] = (pe64 @ Wpe.double().t()).float().contiguous()
        if os.environ.get("SAMPT_DEC_F16X3", "1") != "0" and W.shape[1] % 32 == 0:
            _put_split(out, key + "_w_hl", W)

    for i in range(cfg.dec_depth):
        lp = f"{tr}layers.{i}."
        fuse(lp + "__kvq", [lp + "cross_attn_token_to_image.k_proj", lp + "cross_attn_token_to_image.v_proj",
                            lp + "cross_attn_image_to_token.q_proj"], [True, False, True])
    fuse(tr + "__fin_kv", [tr + "final_attn_token_to_image.k_proj", tr + "final_attn_token_to_image.v_proj"], [True, False])
    out["mask_decoder.__up0_map"] = _shuffle_map(cfg.grid, max_frames)
    out["mask_decoder.__up1_map"] = _shuffle_map(2 * cfg.grid, max_frames)
    if os.environ.get("SAMPT_DEC_F16X3", "1") != "0":
        # split-fp16 planes of the two-way transformer's attention projections: the engine runs the projections over the
        # image tokens (M = frames * grid^2 >= 2048 rows) as 3-term split-fp16 MFMAs, fp32-grade at 2.25x the f32 MFMA rate
        # (measured +1.4 % end to end, profiles/r2_v0_bench_vith_dec_f16x3.log; SAMPT_DEC_F16X3=0 restores the f32 MFMAs)
        for k in [k for k in out if k.startswith("mask_decoder.transformer.") and k.endswith("_proj.weight")]:
            if out[k].shape[1] % 32 == 0 and out[k].shape[0] % 4 == 0:
                _put_split(out, k + "_hl", out[k])
    return {k: v.to(device) for k, v in out.items()}


# ---- fp16 ViT mode: static bias correction of the weight rounding (SamPredictor._select_bias_set) -----------------
VIT_GEMM_KINDS = ("attn.qkv", "attn.proj", "mlp.lin1", "mlp.lin2")      # order of sampt_vit_calibrate's `kind` index


def vit_bias_correction(sd: Dict[str, torch.Tensor], cfg: SamConfig, abar: torch.Tensor) -> Dict[str, torch.Tensor]:
    """b' = b + (W - fp16(W)) . abar for the four GEMMs of every ViT block, in fp64 on the host.

    ``abar`` (depth, 4, >= mlp_ratio * embed_dim): the column means of each GEMM's A operand over a calibration frame's tokens, as
    ``sampt_vit_calibrate`` records them (kind index = ``VIT_GEMM_KINDS``).  ``W - fp16(W)`` is exactly what ``pack_vit``'s ``.half()``
    dropped, so with these biases the token-mean of the fp16 GEMM's output error vanishes for inputs whose column means equal
    ``abar``: mean_t[A.fp16(W)^T + b'] = mean_t[A.W^T + b].  -> {bias name: float32 tensor (CPU)}."""
    D, ld = cfg.embed_dim, cfg.mlp_ratio * cfg.embed_dim
    abar = abar.detach().cpu().double()
    out = {}
    for i in range(cfg.depth):
        for kind, mod in enumerate(VIT_GEMM_KINDS):
            K = ld if mod == "mlp.lin2" else D
            name = f"image_encoder.blocks.{i}.{mod}"
            w = sd[name + ".weight"].detach().float().reshape(-1, K).cpu()
            dw = (w - w.half().float()).double()
            out[name + ".bias"] = (sd[name + ".bias"].detach().cpu().double() + dw @ abar[i, kind, :K]).float()
    return out
