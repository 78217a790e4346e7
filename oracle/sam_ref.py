"""CPU oracle for the SAM half of the SAM-PT hot path.  TEST INFRASTRUCTURE ONLY.

Imported only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.

The reference (SysCV/sam-pt) takes SAM from a third-party pin that is NOT under /root/reference:
``segment_anything`` = facebookresearch/segment-anything @ aac76a1fb03cf90dc7cb2ad481d511642e51aeba
(requirements.txt:26).  This file restates that package's published algorithm (ImageEncoderViT,
PromptEncoder, MaskDecoder/TwoWayTransformer, Sam.preprocess/postprocess_masks, SamPredictor) in
plain PyTorch fp32, driven by the reference's own hyper-parameters
(configs/model/sam/image_encoder/vit_*.yaml, configs/model/sam/{mask_decoder,prompt_encoder}/sam.yaml,
configs/model/sam/sam_vit_base.yaml:10-16) and anchored on the reference's call sites
(sam_pt/modeling/sam_pt.py:771, 783-828, 849).

Parity status: the reference has no tests / golden vectors for this boundary, and
the upstream source is absent, so the pin is a *secondary* one: ``tests/test_oracle_pins.py`` copies the
weights into HuggingFace ``transformers.models.sam`` (an independent implementation of the same
arithmetic that ships in this image) and requires agreement to fp32 round-off on image embeddings,
low-res mask logits and IoU predictions; ``postprocess_masks``, ``get_preprocess_shape`` and the prompt-coordinate transform
are pinned bit for bit on HF's ``SamImageProcessor`` / ``SamProcessor`` (live, no weights involved).  What HF cannot check (the
rest of the SamPredictor plumbing) is marked "parity unpinned" where it occurs.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from sam_pt_amd.weights import SamConfig

SD = Dict[str, torch.Tensor]


def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln2d(x, sd, p, eps=1e-6):
    """LayerNorm2d of segment_anything/modeling/common.py: normalise over the channel dim of NCHW."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + ".weight"][None, :, None, None] * x + sd[p + ".bias"][None, :, None, None]


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# ---------------------------------------------------------------------------------------------
# image encoder
# ---------------------------------------------------------------------------------------------
def _rel_table(rel_pos: torch.Tensor, size: int) -> torch.Tensor:
    """(2*size-1, hd) table -> (size, size, hd) with R[i, j] = rel_pos[i - j + size - 1]."""
    assert rel_pos.shape[0] == 2 * size - 1, "interpolated rel-pos tables are not used by SAM-PT configs"
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + (size - 1)
    return rel_pos[idx]


def vit_attention(sd: SD, p: str, x: torch.Tensor, num_heads: int) -> torch.Tensor:
    """x (B,H,W,D) -> (B,H,W,D); decomposed relative position bias added to the scaled logits."""
    B, H, W, D = x.shape
    hd = D // num_heads
    qkv = _lin(x, sd, p + ".qkv").reshape(B, H * W, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * num_heads, H * W, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    Rh = _rel_table(sd[p + ".rel_pos_h"], H)
    Rw = _rel_table(sd[p + ".rel_pos_w"], W)
    rq = q.reshape(B * num_heads, H, W, hd)                       # un-scaled q
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(-1, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(-1, H * W, H * W)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).view(B, num_heads, H, W, hd).permute(0, 2, 3, 1, 4).reshape(B, H, W, D)
    return _lin(out, sd, p + ".proj")


def _window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def _window_unpartition(w, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def vit_block(sd: SD, cfg: SamConfig, i: int, x: torch.Tensor) -> torch.Tensor:
    p = f"image_encoder.blocks.{i}"
    ws = 0 if i in cfg.global_attn_indexes else cfg.window_size
    shortcut = x
    x = _ln(x, sd, p + ".norm1", 1e-6)
    if ws > 0:
        H, W = x.shape[1:3]
        x, pad_hw = _window_partition(x, ws)          # zero padding AFTER norm1; padded tokens are attended to
    x = vit_attention(sd, p + ".attn", x, cfg.num_heads)
    if ws > 0:
        x = _window_unpartition(x, ws, pad_hw, (H, W))
    x = shortcut + x
    y = _ln(x, sd, p + ".norm2", 1e-6)
    y = _lin(F.gelu(_lin(y, sd, p + ".mlp.lin1")), sd, p + ".mlp.lin2")
    return x + y


def image_encoder(sd: SD, cfg: SamConfig, x: torch.Tensor, trace: Optional[dict] = None, return_interm: bool = False):
    """x (B,3,S,S) normalised+padded -> (B,256,S/16,S/16) [, interm = output (B,g,g,D) of the first global-attention
    block, the only intermediate embedding HQ-SAM uses (App. A-5)]."""
    x = F.conv2d(x, sd["image_encoder.patch_embed.proj.weight"], sd["image_encoder.patch_embed.proj.bias"],
                 stride=cfg.patch_size).permute(0, 2, 3, 1)
    x = x + sd["image_encoder.pos_embed"]
    if trace is not None:
        trace["tokens0"] = x.clone()
    interm = None
    for i in range(cfg.depth):
        x = vit_block(sd, cfg, i, x)
        if trace is not None:
            trace[f"block{i}"] = x.clone()
        if i == cfg.global_attn_indexes[0]:
            interm = x
    tokens = x
    x = x.permute(0, 3, 1, 2)
    x = _ln2d(F.conv2d(x, sd["image_encoder.neck.0.weight"]), sd, "image_encoder.neck.1")
    x = _ln2d(F.conv2d(x, sd["image_encoder.neck.2.weight"], padding=1), sd, "image_encoder.neck.3")
    del tokens
    return (x, interm) if return_interm else x


def preprocess(cfg: SamConfig, x: torch.Tensor) -> torch.Tensor:
    """Sam.preprocess: (x - mean)/std then zero-pad bottom/right to img_size (App. A-2). x (B,3,h,w) float."""
    mean = torch.tensor(cfg.pixel_mean).view(1, 3, 1, 1)
    std = torch.tensor(cfg.pixel_std).view(1, 3, 1, 1)
    x = (x - mean) / std
    h, w = x.shape[-2:]
    return F.pad(x, (0, cfg.img_size - w, 0, cfg.img_size - h))


# ---------------------------------------------------------------------------------------------
# prompt encoder (App. A-4)
# ---------------------------------------------------------------------------------------------
def _pe_encode(sd: SD, coords01: torch.Tensor) -> torch.Tensor:
    c = 2 * coords01 - 1
    c = c @ sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = 2 * math.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd: SD, cfg: SamConfig) -> torch.Tensor:
    """PromptEncoder.get_dense_pe(): (1,256,g,g)."""
    g = cfg.grid
    ones = torch.ones(g, g)
    y = (ones.cumsum(0) - 0.5) / g
    x = (ones.cumsum(1) - 0.5) / g
    return _pe_encode(sd, torch.stack([x, y], dim=-1)).permute(2, 0, 1).unsqueeze(0)


def prompt_encoder(sd: SD, cfg: SamConfig, points: Optional[Tuple[torch.Tensor, torch.Tensor]],
                   boxes: Optional[torch.Tensor], masks: Optional[torch.Tensor]):
    """points=(coords (B,k,2) in input-frame px, labels (B,k)); boxes (B,4); masks (B,1,4g,4g).
    Returns sparse (B,n,256), dense (B,256,g,g)."""
    S = float(cfg.img_size)
    B = 1
    sparse = []
    if points is not None:
        coords, labels = points
        B = coords.shape[0]
        coords = coords + 0.5
        if boxes is None:  # pad with a "not a point"
            coords = torch.cat([coords, torch.zeros(B, 1, 2)], dim=1)
            labels = torch.cat([labels, -torch.ones(B, 1, dtype=labels.dtype)], dim=1)
        pe = _pe_encode(sd, coords / torch.tensor([S, S]))
        pe[labels == -1] = 0.0
        pe[labels == -1] += sd["prompt_encoder.not_a_point_embed.weight"]
        pe[labels == 0] += sd["prompt_encoder.point_embeddings.0.weight"]
        pe[labels == 1] += sd["prompt_encoder.point_embeddings.1.weight"]
        sparse.append(pe)
    if boxes is not None:
        B = boxes.shape[0]
        c = (boxes + 0.5).reshape(-1, 2, 2)
        pe = _pe_encode(sd, c / torch.tensor([S, S]))
        pe[:, 0, :] += sd["prompt_encoder.point_embeddings.2.weight"][0]
        pe[:, 1, :] += sd["prompt_encoder.point_embeddings.3.weight"][0]
        sparse.append(pe)
    sparse = torch.cat(sparse, dim=1) if sparse else torch.zeros(B, 0, cfg.out_chans)
    if masks is not None:
        p = "prompt_encoder.mask_downscaling"
        d = F.conv2d(masks, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2)
        d = F.gelu(_ln2d(d, sd, p + ".1"))
        d = F.conv2d(d, sd[p + ".3.weight"], sd[p + ".3.bias"], stride=2)
        d = F.gelu(_ln2d(d, sd, p + ".4"))
        dense = F.conv2d(d, sd[p + ".6.weight"], sd[p + ".6.bias"])
    else:
        dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, cfg.grid, cfg.grid)
    return sparse, dense


# ---------------------------------------------------------------------------------------------
# mask decoder (App. A-4)
# ---------------------------------------------------------------------------------------------
def _dec_attn(sd: SD, p: str, q, k, v, heads: int):
    q, k, v = _lin(q, sd, p + ".q_proj"), _lin(k, sd, p + ".k_proj"), _lin(v, sd, p + ".v_proj")
    B, nq, C = q.shape
    hd = C // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, hd).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    a = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    o = (a.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, nq, C)
    return _lin(o, sd, p + ".out_proj")


def two_way_transformer(sd: SD, cfg: SamConfig, src: torch.Tensor, pos: torch.Tensor, tokens: torch.Tensor):
    """src,pos (B,256,g,g); tokens (B,n,256) -> (queries (B,n,256), keys (B,g*g,256))."""
    T = "mask_decoder.transformer"
    keys = src.flatten(2).permute(0, 2, 1)
    kpe = pos.flatten(2).permute(0, 2, 1)
    queries, qpe = tokens, tokens
    H = cfg.dec_heads
    for i in range(cfg.dec_depth):
        p = f"{T}.layers.{i}"
        if i == 0:
            queries = _dec_attn(sd, p + ".self_attn", queries, queries, queries, H)
        else:
            q = queries + qpe
            queries = queries + _dec_attn(sd, p + ".self_attn", q, q, queries, H)
        queries = _ln(queries, sd, p + ".norm1", 1e-5)
        q, k = queries + qpe, keys + kpe
        queries = _ln(queries + _dec_attn(sd, p + ".cross_attn_token_to_image", q, k, keys, H), sd, p + ".norm2", 1e-5)
        m = _lin(F.relu(_lin(queries, sd, p + ".mlp.lin1")), sd, p + ".mlp.lin2")
        queries = _ln(queries + m, sd, p + ".norm3", 1e-5)
        q, k = queries + qpe, keys + kpe
        keys = _ln(keys + _dec_attn(sd, p + ".cross_attn_image_to_token", k, q, queries, H), sd, p + ".norm4", 1e-5)
    q, k = queries + qpe, keys + kpe
    queries = _ln(queries + _dec_attn(sd, T + ".final_attn_token_to_image", q, k, keys, H), sd,
                  T + ".norm_final_attn", 1e-5)
    return queries, keys


def _mlp3(sd: SD, p: str, x, n: int):
    for i in range(n):
        x = _lin(x, sd, f"{p}.layers.{i}")
        if i < n - 1:
            x = F.relu(x)
    return x


def hq_features(sd: SD, image_embeddings, interm):
    """HQ-SAM per-image features (1,32,4g,4g) = embedding_encoder(emb) + compress_vit_feat(interm[0]) (App. A-5)."""
    M = "mask_decoder."

    def up(x, p, eps=1e-6):
        x = F.conv_transpose2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2)
        x = F.gelu(_ln2d(x, sd, p + ".1", eps))
        return F.conv_transpose2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], stride=2)

    return up(image_embeddings, M + "embedding_encoder") + up(interm.permute(0, 3, 1, 2), M + "compress_vit_feat")


def mask_decoder(sd: SD, cfg: SamConfig, image_embeddings, image_pe, sparse, dense, multimask_output: bool,
                 hq_feat=None, hq_token_only: bool = False, _hf_samhq_quirk: bool = False):
    """-> (low_res masks (B,m,4g,4g), iou (B,m)).  With ``hq_feat`` (HQ-SAM, MaskDecoderHQ): one extra output token whose
    hyper-network output multiplies embedding_maskfeature(upscaled) + hq_feat; result = SAM mask + HQ mask.

    ``_hf_samhq_quirk`` (pin test only): transformers' SamHQMaskDecoder (modeling_sam_hq.py:993-1003 in this image) drops
    the transformer's updated image keys and upscales the *pre-transformer* embedding with H/W swapped, unlike upstream
    sam-hq (and unlike transformers' own SamMaskDecoder).  The switch reproduces that so every other HQ step can be pinned
    against HF; the keys->upscaling step itself is shared with plain SAM and pinned by sam_hf.npz."""
    nmt = cfg.num_multimask_outputs + 1
    out_tok = torch.cat([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], dim=0)
    if hq_feat is not None:
        out_tok = torch.cat([out_tok, sd["mask_decoder.hf_token.weight"]], dim=0)
    tokens = torch.cat([out_tok.unsqueeze(0).expand(sparse.shape[0], -1, -1), sparse], dim=1)
    src = torch.repeat_interleave(image_embeddings, tokens.shape[0], dim=0) + dense
    pos = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
    b, c, h, w = src.shape
    hs, keys = two_way_transformer(sd, cfg, src, pos, tokens)
    iou_tok, mask_toks = hs[:, 0, :], hs[:, 1:1 + nmt, :]
    if _hf_samhq_quirk:
        src = src.transpose(2, 3).reshape(b, c, h, w)
    else:
        src = keys.transpose(1, 2).reshape(b, c, h, w)
    U = "mask_decoder.output_upscaling"
    up = F.conv_transpose2d(src, sd[U + ".0.weight"], sd[U + ".0.bias"], stride=2)
    up = F.gelu(_ln2d(up, sd, U + ".1"))
    up = F.gelu(F.conv_transpose2d(up, sd[U + ".3.weight"], sd[U + ".3.bias"], stride=2))
    hyper = torch.stack([_mlp3(sd, f"mask_decoder.output_hypernetworks_mlps.{i}", mask_toks[:, i, :], 3)
                         for i in range(nmt)], dim=1)
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
    iou = _mlp3(sd, "mask_decoder.iou_prediction_head", iou_tok, cfg.iou_head_depth)
    sl = slice(1, None) if multimask_output else slice(0, 1)
    if hq_feat is None:
        return masks[:, sl], iou[:, sl]
    assert not multimask_output, "SAM-PT only uses multimask_output=False"
    E = "mask_decoder.embedding_maskfeature"
    uh = F.conv2d(up, sd[E + ".0.weight"], sd[E + ".0.bias"], padding=1)
    uh = F.gelu(_ln2d(uh, sd, E + ".1"))
    uh = F.conv2d(uh, sd[E + ".3.weight"], sd[E + ".3.bias"], padding=1) + hq_feat
    hyper_hq = _mlp3(sd, "mask_decoder.hf_mlp", hs[:, 1 + nmt, :], 3)
    mask_hq = (hyper_hq.unsqueeze(1) @ uh.view(b, c, h * w)).view(b, 1, h, w)
    out = mask_hq if hq_token_only else masks[:, sl] + mask_hq
    return out, iou[:, sl]


def postprocess_masks(cfg: SamConfig, masks, input_size, original_size):
    """Sam.postprocess_masks (App. A-2).  Pinned on HuggingFace's SamImageProcessor.post_process_masks (same three steps, bit-identical:
    tests/test_oracle_pins.py::test_postprocess_and_preprocess_shape_vs_hf)."""
    masks = F.interpolate(masks, (cfg.img_size, cfg.img_size), mode="bilinear", align_corners=False)
    masks = masks[..., :input_size[0], :input_size[1]]
    return F.interpolate(masks, original_size, mode="bilinear", align_corners=False)


# ---------------------------------------------------------------------------------------------
# SamPredictor (App. A-1) — the object the reference drives at sam_pt.py:771, 783-828, 849
# ---------------------------------------------------------------------------------------------
def get_preprocess_shape(oldh: int, oldw: int, long_side: int) -> Tuple[int, int]:
    s = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * s + 0.5), int(oldw * s + 0.5)


class _Transform:
    def __init__(self, target_length):
        self.target_length = target_length

    def apply_coords(self, coords: np.ndarray, original_size) -> np.ndarray:
        oh, ow = original_size
        nh, nw = get_preprocess_shape(oh, ow, self.target_length)
        c = np.array(coords, dtype=float, copy=True)
        c[..., 0] = c[..., 0] * (nw / ow)
        c[..., 1] = c[..., 1] * (nh / oh)
        return c


class _Model:
    mask_threshold = 0.0
    device = torch.device("cpu")


class SamPredictorRef:
    """fp32 CPU stand-in for ``segment_anything.SamPredictor`` over the functional oracle above."""

    def __init__(self, sd: SD, cfg: SamConfig, hq: bool = False):
        self.sd, self.cfg, self.hq = sd, cfg, hq
        self.hq_feat = None
        self.model = _Model()
        self.transform = _Transform(cfg.img_size)
        self.features = None
        self.original_size = self.input_size = None
        self._pe = dense_pe(sd, cfg)
        self.n_set_image = self.n_predict = 0

    @torch.no_grad()
    def set_image(self, image: np.ndarray):
        """image: HxWx3 uint8 RGB.  ResizeLongestSide.apply_image = torchvision resize(to_pil_image(image), target) =
        PIL bilinear (App. A-1); the identity when the longest side is already img_size, which is what the reference
        pipelines feed (configs/demo.yaml:20, vos_eval_root.yaml:28)."""
        H, W = image.shape[:2]
        nh, nw = get_preprocess_shape(H, W, self.cfg.img_size)
        if (nh, nw) != (H, W):
            from PIL import Image
            image = np.array(Image.fromarray(np.ascontiguousarray(image)).resize((nw, nh), Image.BILINEAR))
        x = torch.as_tensor(np.ascontiguousarray(image)).permute(2, 0, 1)[None].float()
        self.original_size, self.input_size = (H, W), (nh, nw)
        if self.hq:
            self.features, interm = image_encoder(self.sd, self.cfg, preprocess(self.cfg, x), return_interm=True)
            self.hq_feat = hq_features(self.sd, self.features, interm)
        else:
            self.features = image_encoder(self.sd, self.cfg, preprocess(self.cfg, x))
        self.n_set_image += 1

    def reset_image(self):
        self.features = self.hq_feat = None
        self.original_size = self.input_size = None

    @torch.no_grad()
    def predict_torch(self, point_coords, point_labels, boxes=None, mask_input=None, multimask_output=True,
                      return_logits=False):
        if self.features is None:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        pts = (point_coords, point_labels) if point_coords is not None else None
        sparse, dense = prompt_encoder(self.sd, self.cfg, pts, boxes.reshape(-1, 4) if boxes is not None else None,
                                       mask_input)
        low, iou = mask_decoder(self.sd, self.cfg, self.features, self._pe, sparse, dense, multimask_output,
                                hq_feat=self.hq_feat if self.hq else None)
        masks = postprocess_masks(self.cfg, low, self.input_size, self.original_size)
        if not return_logits:
            masks = masks > self.model.mask_threshold
        self.n_predict += 1
        return masks, iou, low
