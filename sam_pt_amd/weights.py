"""Seeded random-init state dicts in the *upstream checkpoint key layout*.

No checkpoints exist in this environment, so every model on the hot path is
driven by deterministic random weights.  The key names / shapes follow the real checkpoints so
that a genuine ``.pth`` drops in unchanged:

* PIPS   – ``model-*.pth['model_state_dict']`` of aharley/pips (module tree at
  /root/reference/sam_pt/point_tracker/pips/pips.py:191-287, 290-317, 410-437).
* SAM    – ``sam_vit_{b,l,h}.pth`` of facebookresearch/segment-anything @ aac76a1
  (hyper-parameters at /root/reference/configs/model/sam/**.yaml; key list in SURVEY.md App. C).

The generator is plain tensor code (CPU ``torch.Generator``), identical on every machine with the
same torch build, so the oracle and the HIP path always see the same numbers.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
class _Init:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)
        self.sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def uniform(self, name, shape, bound):
        self.sd[name] = (torch.rand(shape, generator=self.g, dtype=torch.float32) * 2 - 1) * bound

    def normal(self, name, shape, std, mean=0.0):
        self.sd[name] = torch.randn(shape, generator=self.g, dtype=torch.float32) * std + mean

    def linear(self, prefix, out_f, in_f, bias=True):
        b = 1.0 / math.sqrt(in_f)
        self.uniform(prefix + ".weight", (out_f, in_f), b)
        if bias:
            self.uniform(prefix + ".bias", (out_f,), b)

    def conv(self, prefix, out_c, in_c, kh, kw, bias=True, kaiming_fan_out=False):
        fan_in = in_c * kh * kw
        if kaiming_fan_out:  # pips.py:237-239 (kaiming_normal_, fan_out, relu)
            self.normal(prefix + ".weight", (out_c, in_c, kh, kw), math.sqrt(2.0 / (out_c * kh * kw)))
        else:
            self.uniform(prefix + ".weight", (out_c, in_c, kh, kw), 1.0 / math.sqrt(fan_in))
        if bias:
            self.uniform(prefix + ".bias", (out_c,), 1.0 / math.sqrt(fan_in))

    def norm(self, prefix, dim):
        # affine parameters perturbed away from (1, 0) so that parity tests exercise them
        self.normal(prefix + ".weight", (dim,), 0.05, mean=1.0)
        self.normal(prefix + ".bias", (dim,), 0.05)


# --------------------------------------------------------------------------------------
# PIPS
# --------------------------------------------------------------------------------------
PIPS_S = 8
PIPS_LATENT = 128
PIPS_CORR_LEVELS = 4
PIPS_CORR_RADIUS = 3
PIPS_MIXER_DIM = 512
PIPS_MIXER_DEPTH = 12
PIPS_KITCHEN_DIM = PIPS_CORR_LEVELS * (2 * PIPS_CORR_RADIUS + 1) ** 2 + PIPS_LATENT + 64 * 3 + 3  # 519


def init_pips_state_dict(seed: int = 72, S: int = PIPS_S, delta_scale: float = 0.05,
                         vis_bias: float = 2.0) -> "OrderedDict[str, torch.Tensor]":
    """Random PIPS weights, keys as in the reference module tree (pips.py:410-437).

    ``delta_scale`` shrinks the mixer's output head so that the 6-iteration update is contractive,
    like the trained model's.  With an unscaled random head every iteration moves points by several
    pixels and the 968-rad/px sin/cos flow embedding (utils/misc.py:30-55) amplifies fp32 round-off
    ~10x per iteration — even the reference run twice with a different op order diverges by 0.1 px
    after 6 iterations — which would make trajectory parity meaningless.  ``vis_bias`` centres the
    visibility head near the 0.9 linking threshold (pips/tracker.py:111-148) so both linking branches
    are exercised.  Pass ``delta_scale=1.0, vis_bias=0.0`` for an untouched default-style init.
    """
    w = _Init(seed)
    # --- fnet: BasicEncoder (pips.py:191-287), instance norm => no norm parameters
    w.conv("fnet.conv1", 64, 3, 7, 7, kaiming_fan_out=True)
    in_planes = 64
    for li, (dim, stride) in enumerate([(64, 1), (96, 2), (128, 2), (128, 2)], start=1):
        for bi in range(2):
            cin = in_planes if bi == 0 else dim
            p = f"fnet.layer{li}.{bi}"
            w.conv(p + ".conv1", dim, cin, 3, 3, kaiming_fan_out=True)
            w.conv(p + ".conv2", dim, dim, 3, 3, kaiming_fan_out=True)
            if bi == 0 and stride != 1:
                w.conv(p + ".downsample.0", dim, cin, 1, 1, kaiming_fan_out=True)
        in_planes = dim
    w.conv("fnet.conv2", 2 * PIPS_LATENT, 128 + 128 + 96 + 64, 3, 3, kaiming_fan_out=True)
    w.conv("fnet.conv3", PIPS_LATENT, 2 * PIPS_LATENT, 1, 1, kaiming_fan_out=True)
    # --- delta block: MLP-Mixer (pips.py:115-128, 290-317)
    d = PIPS_MIXER_DIM
    w.linear("delta_block.to_delta.0", d, PIPS_KITCHEN_DIM)
    for i in range(1, PIPS_MIXER_DEPTH + 1):
        p = f"delta_block.to_delta.{i}"
        w.norm(p + ".0.norm", d)
        w.conv(p + ".0.fn.0", 4 * S, S, 1, 1)  # Conv1d(k=1): squeeze last dim below
        w.conv(p + ".0.fn.3", S, 4 * S, 1, 1)
        for k in (".0.fn.0.weight", ".0.fn.3.weight"):
            w.sd[p + k] = w.sd[p + k].squeeze(-1)
        w.norm(p + ".1.norm", d)
        w.linear(p + ".1.fn.0", 4 * d, d)
        w.linear(p + ".1.fn.3", d, 4 * d)
    w.norm(f"delta_block.to_delta.{PIPS_MIXER_DEPTH + 1}", d)
    w.linear(f"delta_block.to_delta.{PIPS_MIXER_DEPTH + 3}", S * (PIPS_LATENT + 2), d)
    # --- heads (pips.py:427-437)
    w.norm("norm", PIPS_LATENT)
    w.linear("ffeat_updater.0", PIPS_LATENT, PIPS_LATENT)
    w.linear("vis_predictor.0", 1, PIPS_LATENT)
    head = f"delta_block.to_delta.{PIPS_MIXER_DEPTH + 3}"
    w.sd[head + ".weight"] *= delta_scale
    w.sd[head + ".bias"] *= delta_scale
    w.sd["vis_predictor.0.bias"] += vis_bias
    return w.sd


# --------------------------------------------------------------------------------------
# CoTracker v1 (facebookresearch/co-tracker @ 4f297a9, requirements.txt:29 — third-party, absent from the reference tree)
# --------------------------------------------------------------------------------------
COTRACKER_HIDDEN = 384
COTRACKER_HEADS = 8
COTRACKER_DEPTH = 6          # time_depth = space_depth = 6 (build_cotracker: cotracker_stride_4_wind_8.pth)
COTRACKER_INPUT_DIM = 456    # 130 flow embedding + 196 correlation + 128 track feature + 2 (track mask, visibility)


def init_cotracker_state_dict(seed: int = 72, delta_scale: float = 0.003, vis_bias: float = 0.0,
                              hidden: int = COTRACKER_HIDDEN, depth: int = COTRACKER_DEPTH) -> "OrderedDict[str, torch.Tensor]":
    """Random CoTracker weights with the key layout of ``cotracker_stride_4_wind_8.pth``: ``fnet.*`` =
    the same BasicEncoder tree as PIPS (instance norm: no norm parameters), ``updateformer.input_transform``,
    ``updateformer.{time,space}_blocks.{i}.{attn.qkv, attn.proj, mlp.fc1, mlp.fc2}`` (timm ``Attention`` / ``Mlp``; the
    blocks' LayerNorms are affine-free), ``updateformer.flow_head``, ``norm`` (GroupNorm), ``ffeat_updater.0``,
    ``vis_predictor.0``.  UpdateFormer linears are xavier-uniform with zero bias as upstream's ``initialize_weights``.
    ``delta_scale``: see ``init_pips_state_dict`` — with xavier weights and 12 residual blocks the unscaled random
    head moves points by pixels per iteration and the 968-rad/px flow embedding makes the model chaotic (a 1e-7 relative
    weight perturbation moves trajectories by 2.3 px at 0.05, 0.34 px at 0.01, 6e-4 px at 0.003), so 0.003 is the
    default.  The random visibility head already straddles the 0.7 threshold of configs/model/point_tracker/
    cotracker.yaml:7 (logits 1.6 +- 0.8 vs logit(0.7) = 0.85: ~80 % visible), so ``vis_bias`` defaults to 0."""
    w = _Init(seed)
    w.conv("fnet.conv1", 64, 3, 7, 7, kaiming_fan_out=True)
    in_planes = 64
    for li, (dim, stride) in enumerate([(64, 1), (96, 2), (128, 2), (128, 2)], start=1):
        for bi in range(2):
            cin = in_planes if bi == 0 else dim
            p = f"fnet.layer{li}.{bi}"
            w.conv(p + ".conv1", dim, cin, 3, 3, kaiming_fan_out=True)
            w.conv(p + ".conv2", dim, dim, 3, 3, kaiming_fan_out=True)
            if bi == 0 and stride != 1:
                w.conv(p + ".downsample.0", dim, cin, 1, 1, kaiming_fan_out=True)
        in_planes = dim
    w.conv("fnet.conv2", 2 * PIPS_LATENT, 128 + 128 + 96 + 64, 3, 3, kaiming_fan_out=True)
    w.conv("fnet.conv3", PIPS_LATENT, 2 * PIPS_LATENT, 1, 1, kaiming_fan_out=True)

    def xavier(prefix, out_f, in_f):
        w.uniform(prefix + ".weight", (out_f, in_f), math.sqrt(6.0 / (in_f + out_f)))
        w.sd[prefix + ".bias"] = torch.zeros(out_f)

    xavier("updateformer.input_transform", hidden, COTRACKER_INPUT_DIM)
    xavier("updateformer.flow_head", PIPS_LATENT + 2, hidden)
    for kind in ("time_blocks", "space_blocks"):
        for i in range(depth):
            p = f"updateformer.{kind}.{i}"
            xavier(p + ".attn.qkv", 3 * hidden, hidden)
            xavier(p + ".attn.proj", hidden, hidden)
            xavier(p + ".mlp.fc1", 4 * hidden, hidden)
            xavier(p + ".mlp.fc2", hidden, 4 * hidden)
    w.norm("norm", PIPS_LATENT)
    w.linear("ffeat_updater.0", PIPS_LATENT, PIPS_LATENT)
    w.linear("vis_predictor.0", 1, PIPS_LATENT)
    w.sd["updateformer.flow_head.weight"] *= delta_scale
    w.sd["updateformer.flow_head.bias"] *= delta_scale
    w.sd["vis_predictor.0.bias"] += vis_bias
    return w.sd


PIPS2_KITCHEN = 3 * PIPS_CORR_LEVELS * (2 * PIPS_CORR_RADIUS + 1) ** 2 + PIPS_LATENT + 2      # 718
PIPS2_BLOCKS = [(128, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512), (512, 1024), (1024, 1024)]


def init_pips2_state_dict(seed: int = 72, delta_scale: float = 0.05) -> "OrderedDict[str, torch.Tensor]":
    """Random PIPS++ weights, keys as in the reference module tree (pips_plus_plus.py:420-434: ``fnet`` BasicEncoder
    (:180-260, instance norm -> no norm parameters), ``delta_block`` 1-D ResNet (:263-342) and the unused ``norm``).
    ``delta_scale`` shrinks the final dense layer so that the 16-iteration update is contractive (see
    ``init_pips_state_dict``).  Unused-but-present modules of the reference (``first_block_norm``, ``final_norm``,
    InstanceNorm without affine) have no parameters; ``norm`` (GroupNorm) has and is kept so the dict strict-loads."""
    w = _Init(seed)
    w.conv("fnet.conv1", 64, 3, 7, 7, kaiming_fan_out=True)
    in_planes = 64
    for li, (dim, stride) in enumerate([(64, 1), (96, 2), (128, 2), (128, 2)], start=1):
        for bi in range(2):
            cin = in_planes if bi == 0 else dim
            p = f"fnet.layer{li}.{bi}"
            w.conv(p + ".conv1", dim, cin, 3, 3, kaiming_fan_out=True)
            w.conv(p + ".conv2", dim, dim, 3, 3, kaiming_fan_out=True)
            if bi == 0 and stride != 1:
                w.conv(p + ".downsample.0", dim, cin, 1, 1, kaiming_fan_out=True)
        in_planes = dim
    w.conv("fnet.conv2", 2 * PIPS_LATENT, 128 + 128 + 96 + 64, 3, 3, kaiming_fan_out=True)
    w.conv("fnet.conv3", PIPS_LATENT, 2 * PIPS_LATENT, 1, 1, kaiming_fan_out=True)

    def conv1d(name, cout, cin):
        w.conv(name, cout, cin, 3, 1)
        w.sd[name + ".weight"] = w.sd[name + ".weight"].squeeze(-1)          # (cout, cin, 3)

    conv1d("delta_block.first_block_conv.conv", 128, PIPS2_KITCHEN)
    for i, (cin, cout) in enumerate(PIPS2_BLOCKS):
        conv1d(f"delta_block.basicblock_list.{i}.conv1.conv", cout, cin)
        conv1d(f"delta_block.basicblock_list.{i}.conv2.conv", cout, cout)
    w.linear("delta_block.dense", 2, 1024)
    w.sd["delta_block.dense.weight"] *= delta_scale
    w.sd["delta_block.dense.bias"] *= delta_scale
    w.norm("norm", PIPS_LATENT)
    return w.sd


# --------------------------------------------------------------------------------------
# SAM
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class SamConfig:
    """Hyper-parameters of one SAM variant (configs/model/sam/image_encoder/vit_*.yaml,
    configs/model/sam/{mask_decoder,prompt_encoder}/sam.yaml, sam_vit_base.yaml:10-16)."""
    name: str
    embed_dim: int
    depth: int
    num_heads: int
    global_attn_indexes: Tuple[int, ...]
    img_size: int = 1024
    patch_size: int = 16
    window_size: int = 14
    mlp_ratio: int = 4
    out_chans: int = 256          # prompt_embed_dim
    mask_in_chans: int = 16
    dec_depth: int = 2
    dec_heads: int = 8
    dec_mlp_dim: int = 2048
    num_multimask_outputs: int = 3
    iou_head_depth: int = 3
    iou_head_hidden_dim: int = 256
    pixel_mean: Tuple[float, float, float] = (123.675, 116.28, 103.53)
    pixel_std: Tuple[float, float, float] = (58.395, 57.12, 57.375)

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


SAM_CONFIGS: Dict[str, SamConfig] = {
    "vit_b": SamConfig("vit_b", 768, 12, 12, (2, 5, 8, 11)),
    "vit_l": SamConfig("vit_l", 1024, 24, 16, (5, 11, 17, 23)),
    "vit_h": SamConfig("vit_h", 1280, 32, 16, (7, 15, 23, 31)),
    # reduced geometry for fast CPU tests only (same code paths: windows, padding, global blocks)
    "vit_test": SamConfig("vit_test", 64, 2, 2, (1,), img_size=256, patch_size=16, window_size=6,
                          out_chans=256),
}


def init_sam_state_dict(cfg: SamConfig, seed: int = 72, hq: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Random SAM weights with the upstream ``sam_vit_*.pth`` key layout; ``hq=True`` adds the HQ-SAM decoder extras of
    ``sam_hq_vit_*.pth`` (m43/sam-hq @ 75c73fa, MaskDecoderHQ: SURVEY.md App. A-5)."""
    w = _Init(seed)
    D, g, hd = cfg.embed_dim, cfg.grid, cfg.head_dim
    C = cfg.out_chans
    # ---- image encoder
    w.conv("image_encoder.patch_embed.proj", D, 3, cfg.patch_size, cfg.patch_size)
    w.normal("image_encoder.pos_embed", (1, g, g, D), 0.02)
    for i in range(cfg.depth):
        p = f"image_encoder.blocks.{i}"
        s = g if i in cfg.global_attn_indexes else cfg.window_size
        w.norm(p + ".norm1", D)
        w.linear(p + ".attn.qkv", 3 * D, D)
        w.linear(p + ".attn.proj", D, D)
        w.normal(p + ".attn.rel_pos_h", (2 * s - 1, hd), 0.05)
        w.normal(p + ".attn.rel_pos_w", (2 * s - 1, hd), 0.05)
        w.norm(p + ".norm2", D)
        w.linear(p + ".mlp.lin1", cfg.mlp_ratio * D, D)
        w.linear(p + ".mlp.lin2", D, cfg.mlp_ratio * D)
    w.conv("image_encoder.neck.0", C, D, 1, 1, bias=False)
    w.norm("image_encoder.neck.1", C)
    w.conv("image_encoder.neck.2", C, C, 3, 3, bias=False)
    w.norm("image_encoder.neck.3", C)
    # ---- prompt encoder
    w.normal("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", (2, C // 2), 1.0)
    for i in range(4):
        w.normal(f"prompt_encoder.point_embeddings.{i}.weight", (1, C), 1.0)
    w.normal("prompt_encoder.not_a_point_embed.weight", (1, C), 1.0)
    w.normal("prompt_encoder.no_mask_embed.weight", (1, C), 1.0)
    m = cfg.mask_in_chans
    w.conv("prompt_encoder.mask_downscaling.0", m // 4, 1, 2, 2)
    w.norm("prompt_encoder.mask_downscaling.1", m // 4)
    w.conv("prompt_encoder.mask_downscaling.3", m, m // 4, 2, 2)
    w.norm("prompt_encoder.mask_downscaling.4", m)
    w.conv("prompt_encoder.mask_downscaling.6", C, m, 1, 1)
    # ---- mask decoder
    T = "mask_decoder.transformer"

    def attn(prefix, downsample):
        inner = C // downsample
        w.linear(prefix + ".q_proj", inner, C)
        w.linear(prefix + ".k_proj", inner, C)
        w.linear(prefix + ".v_proj", inner, C)
        w.linear(prefix + ".out_proj", C, inner)

    for i in range(cfg.dec_depth):
        p = f"{T}.layers.{i}"
        attn(p + ".self_attn", 1)
        w.norm(p + ".norm1", C)
        attn(p + ".cross_attn_token_to_image", 2)
        w.norm(p + ".norm2", C)
        w.linear(p + ".mlp.lin1", cfg.dec_mlp_dim, C)
        w.linear(p + ".mlp.lin2", C, cfg.dec_mlp_dim)
        w.norm(p + ".norm3", C)
        w.norm(p + ".norm4", C)
        attn(p + ".cross_attn_image_to_token", 2)
    attn(T + ".final_attn_token_to_image", 2)
    w.norm(T + ".norm_final_attn", C)
    nmt = cfg.num_multimask_outputs + 1
    w.normal("mask_decoder.iou_token.weight", (1, C), 1.0)
    w.normal("mask_decoder.mask_tokens.weight", (nmt, C), 1.0)
    # ConvTranspose2d weights are (in, out, kh, kw)
    w.uniform("mask_decoder.output_upscaling.0.weight", (C, C // 4, 2, 2), 1.0 / math.sqrt(C))
    w.uniform("mask_decoder.output_upscaling.0.bias", (C // 4,), 1.0 / math.sqrt(C))
    w.norm("mask_decoder.output_upscaling.1", C // 4)
    w.uniform("mask_decoder.output_upscaling.3.weight", (C // 4, C // 8, 2, 2), 1.0 / math.sqrt(C // 4))
    w.uniform("mask_decoder.output_upscaling.3.bias", (C // 8,), 1.0 / math.sqrt(C // 4))
    for i in range(nmt):
        p = f"mask_decoder.output_hypernetworks_mlps.{i}.layers"
        w.linear(p + ".0", C, C)
        w.linear(p + ".1", C, C)
        w.linear(p + ".2", C // 8, C)
    p = "mask_decoder.iou_prediction_head.layers"
    h = cfg.iou_head_hidden_dim
    dims = [C] + [h] * (cfg.iou_head_depth - 1) + [nmt]
    for i in range(cfg.iou_head_depth):
        w.linear(f"{p}.{i}", dims[i + 1], dims[i])
    if hq:
        M = "mask_decoder."
        w.normal(M + "hf_token.weight", (1, C), 1.0)
        w.linear(M + "hf_mlp.layers.0", C, C)
        w.linear(M + "hf_mlp.layers.1", C, C)
        w.linear(M + "hf_mlp.layers.2", C // 8, C)

        def convt(name, cin, cout):
            w.uniform(name + ".weight", (cin, cout, 2, 2), 1.0 / math.sqrt(cin))
            w.uniform(name + ".bias", (cout,), 1.0 / math.sqrt(cin))

        convt(M + "compress_vit_feat.0", D, C)
        w.norm(M + "compress_vit_feat.1", C)
        convt(M + "compress_vit_feat.3", C, C // 8)
        convt(M + "embedding_encoder.0", C, C // 4)
        w.norm(M + "embedding_encoder.1", C // 4)
        convt(M + "embedding_encoder.3", C // 4, C // 8)
        w.conv(M + "embedding_maskfeature.0", C // 4, C // 8, 3, 3)
        w.norm(M + "embedding_maskfeature.1", C // 4)
        w.conv(M + "embedding_maskfeature.3", C // 8, C // 4, 3, 3)
    return w.sd
