"""Secondary pin for the SAM oracle: HuggingFace ``transformers.models.sam`` with copied weights.

TEST INFRASTRUCTURE ONLY.  HF's SamModel is an independent re-implementation of the same arithmetic
as facebookresearch/segment-anything (the reference's third-party pin, absent here).  This module
maps an upstream-layout state dict onto HF parameter names so that the oracle in ``sam_ref.py`` can be
checked number-for-number (SURVEY.md §8c "secondary cross-check").
"""
from __future__ import annotations

import re
from typing import Dict

import torch

from sam_pt_amd.weights import SamConfig


def build_hf_model(cfg: SamConfig, sd: Dict[str, torch.Tensor]):
    from transformers import SamConfig as HC, SamModel
    from transformers.models.sam.configuration_sam import (SamMaskDecoderConfig, SamPromptEncoderConfig,
                                                           SamVisionConfig)
    vc = SamVisionConfig(hidden_size=cfg.embed_dim, output_channels=cfg.out_chans, num_hidden_layers=cfg.depth,
                         num_attention_heads=cfg.num_heads, image_size=cfg.img_size, patch_size=cfg.patch_size,
                         window_size=cfg.window_size, global_attn_indexes=list(cfg.global_attn_indexes),
                         mlp_dim=cfg.mlp_ratio * cfg.embed_dim, layer_norm_eps=1e-6, attn_implementation="eager")
    pc = SamPromptEncoderConfig(hidden_size=cfg.out_chans, image_size=cfg.img_size, patch_size=cfg.patch_size,
                                mask_input_channels=cfg.mask_in_chans)
    mc = SamMaskDecoderConfig(hidden_size=cfg.out_chans, mlp_dim=cfg.dec_mlp_dim, num_hidden_layers=cfg.dec_depth,
                              num_attention_heads=cfg.dec_heads, iou_head_depth=cfg.iou_head_depth,
                              iou_head_hidden_dim=cfg.iou_head_hidden_dim,
                              num_multimask_outputs=cfg.num_multimask_outputs)
    model = SamModel(HC(vision_config=vc, prompt_encoder_config=pc, mask_decoder_config=mc)).eval()
    hf = {}

    def put(dst, src):
        hf[dst] = sd[src].clone()

    g = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    hf["shared_image_embedding.positional_embedding"] = g.clone()
    hf["prompt_encoder.shared_embedding.positional_embedding"] = g.clone()
    for k in sd:
        if k.startswith("image_encoder."):
            d = k.replace("image_encoder.", "vision_encoder.")
            d = d.replace("patch_embed.proj.", "patch_embed.projection.")
            d = d.replace(".blocks.", ".layers.").replace(".norm1.", ".layer_norm1.").replace(".norm2.", ".layer_norm2.")
            d = d.replace("neck.0.", "neck.conv1.").replace("neck.1.", "neck.layer_norm1.")
            d = d.replace("neck.2.", "neck.conv2.").replace("neck.3.", "neck.layer_norm2.")
            put(d, k)
        elif k.startswith("prompt_encoder."):
            if "pe_layer" in k:
                continue
            d = k.replace("point_embeddings.", "point_embed.")
            d = d.replace("mask_downscaling.0.", "mask_embed.conv1.").replace("mask_downscaling.1.", "mask_embed.layer_norm1.")
            d = d.replace("mask_downscaling.3.", "mask_embed.conv2.").replace("mask_downscaling.4.", "mask_embed.layer_norm2.")
            d = d.replace("mask_downscaling.6.", "mask_embed.conv3.")
            put(d, k)
        elif k.startswith("mask_decoder."):
            d = k
            d = re.sub(r"\.norm([1-4])\.", r".layer_norm\1.", d)
            d = d.replace("norm_final_attn", "layer_norm_final_attn")
            d = d.replace("output_upscaling.0.", "upscale_conv1.").replace("output_upscaling.1.", "upscale_layer_norm.")
            d = d.replace("output_upscaling.3.", "upscale_conv2.")
            m = re.match(r"(mask_decoder\.(?:output_hypernetworks_mlps\.\d+|iou_prediction_head))\.layers\.(\d+)\.(\w+)", d)
            if m:
                n = 3  # both MLPs are 3 layers deep in every SAM config
                i = int(m.group(2))
                name = "proj_in" if i == 0 else ("proj_out" if i == n - 1 else f"layers.{i - 1}")
                d = f"{m.group(1)}.{name}.{m.group(3)}"
            put(d, k)
    missing, unexpected = model.load_state_dict(hf, strict=True), None
    return model


@torch.no_grad()
def hf_embed(model, pixel_values: torch.Tensor) -> torch.Tensor:
    return model.get_image_embeddings(pixel_values)


@torch.no_grad()
def hf_decode(model, image_embeddings, points=None, labels=None, boxes=None, masks=None, multimask_output=False):
    """points (1,k,2) input-frame px; labels (1,k); boxes (1,4).  Returns (low_res (1,m,H,W), iou (1,m))."""
    kw = {}
    if points is not None:
        kw["input_points"] = points[:, None]         # (B, point_batch=1, k, 2)
        kw["input_labels"] = labels[:, None].long()
    if boxes is not None:
        kw["input_boxes"] = boxes[:, None].reshape(1, 1, 4)
    if masks is not None:
        kw["input_masks"] = masks
    out = model(image_embeddings=image_embeddings, multimask_output=multimask_output, **kw)
    return out.pred_masks[:, 0], out.iou_scores[:, 0]


def build_hf_hq_model(cfg: SamConfig, sd: Dict[str, torch.Tensor]):
    """HuggingFace ``SamHQModel`` carrying the same weights (HQ-SAM = SAM + MaskDecoderHQ extras, App. A-5)."""
    from transformers.models.sam_hq.configuration_sam_hq import (SamHQConfig, SamHQMaskDecoderConfig,
                                                                 SamHQPromptEncoderConfig, SamHQVisionConfig)
    from transformers.models.sam_hq.modeling_sam_hq import SamHQModel
    vc = SamHQVisionConfig(hidden_size=cfg.embed_dim, output_channels=cfg.out_chans, num_hidden_layers=cfg.depth,
                           num_attention_heads=cfg.num_heads, image_size=cfg.img_size, patch_size=cfg.patch_size,
                           window_size=cfg.window_size, global_attn_indexes=list(cfg.global_attn_indexes),
                           mlp_dim=cfg.mlp_ratio * cfg.embed_dim, layer_norm_eps=1e-6, attn_implementation="eager")
    pc = SamHQPromptEncoderConfig(hidden_size=cfg.out_chans, image_size=cfg.img_size, patch_size=cfg.patch_size,
                                  mask_input_channels=cfg.mask_in_chans)
    mc = SamHQMaskDecoderConfig(hidden_size=cfg.out_chans, mlp_dim=cfg.dec_mlp_dim, num_hidden_layers=cfg.dec_depth,
                                num_attention_heads=cfg.dec_heads, iou_head_depth=cfg.iou_head_depth,
                                iou_head_hidden_dim=cfg.iou_head_hidden_dim,
                                num_multimask_outputs=cfg.num_multimask_outputs, vit_dim=cfg.embed_dim)
    model = SamHQModel(SamHQConfig(vision_config=vc, prompt_encoder_config=pc, mask_decoder_config=mc)).eval()
    hq_keys = ("hf_token", "hf_mlp", "compress_vit_feat", "embedding_encoder", "embedding_maskfeature")
    base = build_hf_model(cfg, {k: v for k, v in sd.items() if not any(h in k for h in hq_keys)})   # SAM key mapping
    hf = {k: v.clone() for k, v in base.state_dict().items()}
    M = "mask_decoder."
    hf[M + "hq_token.weight"] = sd[M + "hf_token.weight"].clone()
    for i, name in enumerate(["proj_in", "layers.0", "proj_out"]):
        hf[f"{M}hq_mask_mlp.{name}.weight"] = sd[f"{M}hf_mlp.layers.{i}.weight"].clone()
        hf[f"{M}hq_mask_mlp.{name}.bias"] = sd[f"{M}hf_mlp.layers.{i}.bias"].clone()
    for src, dst in [("compress_vit_feat", ("compress_vit_conv1", "compress_vit_norm", "compress_vit_conv2")),
                     ("embedding_encoder", ("encoder_conv1", "encoder_norm", "encoder_conv2")),
                     ("embedding_maskfeature", ("mask_conv1", "mask_norm", "mask_conv2"))]:
        for idx, d in zip((0, 1, 3), dst):
            hf[f"{M}{d}.weight"] = sd[f"{M}{src}.{idx}.weight"].clone()
            hf[f"{M}{d}.bias"] = sd[f"{M}{src}.{idx}.bias"].clone()
    model.load_state_dict(hf, strict=True)
    return model


@torch.no_grad()
def hf_hq_embed(model, pixel_values):
    out = model.vision_encoder(pixel_values)
    return out.last_hidden_state, out.intermediate_embeddings


@torch.no_grad()
def hf_hq_decode(model, image_embeddings, interm, points=None, labels=None, boxes=None, masks=None):
    kw = {}
    if points is not None:
        kw["input_points"] = points[:, None]
        kw["input_labels"] = labels[:, None].long()
    if boxes is not None:
        kw["input_boxes"] = boxes[:, None].reshape(1, 1, 4)
    if masks is not None:
        kw["input_masks"] = masks
    out = model(image_embeddings=image_embeddings, intermediate_embeddings=interm, multimask_output=False,
                hq_token_only=False, **kw)
    return out.pred_masks[:, 0], out.iou_scores[:, 0]
