"""Seam 2 of the drop-in boundary: a ``segment_anything.SamPredictor``-compatible object backed by the HIP library.

What the reference touches on a predictor and therefore what is provided, with the same names:
``.model`` (``.device``, ``.mask_threshold``), ``.transform.apply_coords`` / ``apply_coords_torch``,
``.original_size`` / ``.input_size`` / ``.features`` / ``.is_image_set``, ``set_image(np HxWx3 uint8)``,
``predict_torch(point_coords, point_labels, boxes, mask_input, multimask_output, return_logits)`` and the numpy
``predict``.  Call sites: sam_pt/modeling/sam_pt.py:771, 783-828, 849; sam_pt/vos_eval/eval.py:243-250.

Beyond the reference API (used by our own ``SamPt`` for speed, legal inside the same seam — SURVEY.md §7.1):
``encode_frames`` (batched image encoding of a whole clip, embeddings stay in HBM) and ``track_decode`` (the whole
per-(frame, object) prompt chain of ``SamPt.predict_mask`` on device without host syncs).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib
from .pack import VIT_GEMM_KINDS, pack_decoder, pack_vit, vit_bias_correction
from .weights import SAM_CONFIGS, SamConfig, init_sam_state_dict


def pil_bilinear_tables(in_size: int, out_size: int):
    """PIL's fixed-point bilinear resampling tables for one axis (src/libImaging/Resample.c: precompute_coeffs with the
    bilinear filter, support 1 scaled by max(1, in/out), then normalize_coeffs_8bpc with 22 fractional bits).
    -> (coef int32 [out][ksize], bounds int32 [out][2] = (first input index, tap count)).  Bit-exact against PIL 12.2
    (tests/test_cpu_host.py)."""
    import math
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.float64)
    bounds = np.zeros((out_size, 2), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax)
        a = np.abs((x + xmin - center + 0.5) * (1.0 / fs))
        w = np.where(a < 1.0, 1.0 - a, 0.0)
        ww = w.sum()
        kk[xx, :xmax] = w / ww if ww != 0 else w
        bounds[xx] = (xmin, xmax)
    coef = np.trunc(np.where(kk < 0, -0.5 + kk * (1 << 22), 0.5 + kk * (1 << 22))).astype(np.int32)
    return coef, bounds


class ResizeLongestSide:
    """segment_anything.utils.transforms.ResizeLongestSide (App. A-1): coordinates on the host, ``apply_image`` =
    torchvision ``resize(to_pil_image(image), target)`` = PIL bilinear, done on the device with PIL's own fixed-point
    arithmetic (bit-identical to ``PIL.Image.resize``)."""

    def __init__(self, target_length: int):
        self.target_length = target_length
        self._tables = {}

    @_lib.on_device(lambda self, image, chw=False: image.device)
    def apply_image_torch(self, image: torch.Tensor, chw: bool = False) -> torch.Tensor:
        """uint8 frames on the HIP device, (...,H,W,3) or with ``chw`` (...,3,H,W) -> the same layout with the longest side
        = target_length (identity if it already is)."""
        H, W = (image.shape[-2], image.shape[-1]) if chw else (image.shape[-3], image.shape[-2])
        nh, nw = self.get_preprocess_shape(H, W, self.target_length)
        if (nh, nw) == (H, W):
            return image
        lib, dev = _lib.load(), image.device
        cur = image.contiguous()
        for horizontal, n_in, n_out in ((True, W, nw), (False, H, nh)):    # PIL: horizontal pass first, then vertical
            if n_in == n_out:
                continue
            key = (n_in, n_out, str(dev))
            if key not in self._tables:
                coef, bounds = pil_bilinear_tables(n_in, n_out)
                self._tables[key] = (torch.from_numpy(coef).to(dev), torch.from_numpy(bounds).to(dev), coef.shape[1])
            coef_d, bounds_d, ks = self._tables[key]
            shp = list(cur.shape)
            ax = (len(shp) - 1 if horizontal else len(shp) - 2) if chw else (len(shp) - 2 if horizontal else len(shp) - 3)
            outer = int(np.prod(shp[:ax])) if ax > 0 else 1
            inner = int(np.prod(shp[ax + 1:])) if ax + 1 < len(shp) else 1
            shp[ax] = n_out
            dst = torch.empty(shp, dtype=torch.uint8, device=dev)
            _lib.check(lib.sampt_pil_resample_u8(_lib.ptr(cur), _lib.ptr(dst), outer, n_in, n_out, inner, _lib.ptr(coef_d),
                                                 _lib.ptr(bounds_d), ks, _lib.stream_ptr()), "sampt_pil_resample_u8")
            cur = dst
        return cur

    @staticmethod
    def get_preprocess_shape(oldh: int, oldw: int, long_side_length: int) -> Tuple[int, int]:
        scale = long_side_length * 1.0 / max(oldh, oldw)
        return int(oldh * scale + 0.5), int(oldw * scale + 0.5)

    def apply_coords(self, coords: np.ndarray, original_size) -> np.ndarray:
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        coords = np.array(coords, dtype=float, copy=True)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_coords_torch(self, coords: torch.Tensor, original_size) -> torch.Tensor:
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        coords = coords.clone().to(torch.float)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes(self, boxes: np.ndarray, original_size) -> np.ndarray:
        return self.apply_coords(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)


class ClipFeatures:
    """Per-frame SAM embeddings of a clip plus (HQ-SAM) the per-frame HQ features; indexes/flips along the frame axis like
    the plain embedding tensor it replaces in ``SamPt``."""

    def __init__(self, emb: torch.Tensor, hq: torch.Tensor):
        self.emb, self.hq = emb, hq

    def __len__(self):
        return self.emb.shape[0]

    @property
    def shape(self):
        return self.emb.shape

    def __getitem__(self, i):
        return ClipFeatures(self.emb[i], self.hq[i])

    def flip(self, *dims):
        return ClipFeatures(self.emb.flip(*dims), self.hq.flip(*dims))

    def index_select(self, dim, idx):
        return ClipFeatures(self.emb.index_select(dim, idx), self.hq.index_select(dim, idx))


class SamHip(nn.Module):
    """The ``Sam``-like object a predictor exposes as ``.model`` (sam_pt.py:96, 118-120, 334).  Holds the upstream-layout
    state dict; the constructor keywords mirror ``BaseHydra`` (sam_pt/modeling/sam.py:18-31)."""

    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, variant: Optional[str] = None, checkpoint: Optional[str] = None, state_dict=None, seed: int = 72,
                 precision: str = "f16", config: Optional[SamConfig] = None, max_batch: int = 8,
                 max_decode_batch: int = 128, hq: Optional[bool] = None, image_encoder=None, prompt_encoder=None,
                 mask_decoder=None, pixel_mean=None, pixel_std=None, **hydra_kwargs):
        """``hq``: build the HQ-SAM decoder (sam_pt/modeling/sam.py SamHQHydra, configs/model/sam/samhq_vit_*.yaml);
        default: inferred from the checkpoint (presence of ``mask_decoder.hf_token.weight``) or from the ``mask_decoder``
        node's ``_target_`` (``...MaskDecoderHQ``).

        Hydra: this class stands in for ``sam_pt.modeling.sam.Sam*Hydra`` (sam.py:18-61) and takes the very keywords of
        ``configs/model/sam/*.yaml``.  The geometry comes from the nested ``image_encoder`` node (``embed_dim, depth,
        num_heads, global_attn_indexes, window_size, ...`` of configs/model/sam/image_encoder/vit_*.yaml) — a mapping when
        the node is left un-instantiated (``+...sam_model._recursive_=false``), or upstream's
        ``ImageEncoderViT`` object if Hydra did instantiate it; ``variant`` is only a shorthand for code that builds the
        object by hand.  ``prompt_encoder`` / ``mask_decoder`` hyper-parameters are the fixed ones of
        configs/model/sam/{prompt_encoder,mask_decoder}/sam.yaml and are checked, not used."""
        super().__init__()
        if config is None:
            config = self._config_from_hydra(image_encoder, mask_decoder, pixel_mean, pixel_std, hydra_kwargs)
        if config is None:
            config = SAM_CONFIGS[variant if variant is not None else "vit_h"]
        elif variant is not None and (SAM_CONFIGS[variant].embed_dim, SAM_CONFIGS[variant].depth) != (config.embed_dim, config.depth):
            raise ValueError(f"variant={variant} contradicts the image_encoder config (embed_dim {config.embed_dim}, depth {config.depth})")
        self.cfg = config
        if hq is None and mask_decoder is not None:
            tgt = mask_decoder.get("_target_", "") if hasattr(mask_decoder, "get") else type(mask_decoder).__name__
            if str(tgt).endswith("MaskDecoderHQ"):
                hq = True if (state_dict is None and checkpoint is None) else None   # a checkpoint decides by its keys
        if state_dict is None and checkpoint is not None:
            with open(checkpoint, "rb") as f:
                state_dict = torch.load(f, map_location="cpu")
        # no checkpoint: seeded random weights in the upstream key layout, generated on first use (ViT-H is 2.5 GB of fp32)
        self._sd, self._seed = state_dict, seed
        has_hq = bool(hq) if state_dict is None else "mask_decoder.hf_token.weight" in state_dict
        self.hq = has_hq if hq is None else bool(hq)
        if self.hq and not has_hq:
            raise ValueError("hq=True but the checkpoint has no MaskDecoderHQ weights (mask_decoder.hf_token.weight ...)")
        # "f16": fp16 MFMA inputs in the ViT blocks (fast mode); "f16x3": every ViT product from split-fp16 pieces (three fp16
        # MFMAs, fp32 accumulate) — the reference's fp32 arithmetic (sam_pt.py:849) at fp32 grade on the fp16 matrix pipe;
        # "f32": exact f32 MFMAs and materialised attention scores (parity reference, 1/16 of the fp16 MFMA rate)
        assert precision in ("f16", "f16x3", "f32")
        self.precision = precision
        self.max_batch = max_batch
        self.max_decode_batch = max_decode_batch
        self.register_buffer("pixel_mean", torch.tensor(self.cfg.pixel_mean).view(-1, 1, 1), persistent=False)
        self.prompt_embed_dim, self.image_size = self.cfg.out_chans, self.cfg.img_size
        self.vit_patch_size, self.image_embedding_size = self.cfg.patch_size, self.cfg.grid

    @property
    def sd(self):
        if self._sd is None:
            self._sd = init_sam_state_dict(self.cfg, self._seed, hq=self.hq)
        return self._sd

    @staticmethod
    def _config_from_hydra(image_encoder, mask_decoder, pixel_mean, pixel_std, kw) -> Optional[SamConfig]:
        """SamConfig from the keyword arguments Hydra passes for configs/model/sam/*.yaml (None if there is no
        ``image_encoder`` node to read)."""
        if image_encoder is None:
            return None
        if hasattr(image_encoder, "get"):                                   # un-instantiated node (mapping / DictConfig)
            ie = image_encoder
            g = lambda k, d=None: ie.get(k, d)
            embed_dim, depth, heads = int(g("embed_dim")), int(g("depth")), int(g("num_heads"))
            gidx = tuple(int(i) for i in g("global_attn_indexes"))
            window, mlp_ratio = int(g("window_size", 14)), int(g("mlp_ratio", 4))
            img = int(g("img_size", kw.get("image_size", 1024)))
            patch = int(g("patch_size", kw.get("vit_patch_size", 16)))
            out_chans = int(g("out_chans", kw.get("prompt_embed_dim", 256)))
            if not (g("use_rel_pos", True) and g("qkv_bias", True)):
                raise ValueError("SamHip: the HIP encoder implements SAM's ViT (use_rel_pos and qkv_bias are always on)")
        else:                                                               # upstream ImageEncoderViT instance
            blocks = list(image_encoder.blocks)
            embed_dim, depth = int(image_encoder.pos_embed.shape[-1]), len(blocks)
            heads = int(blocks[0].attn.num_heads)
            gidx = tuple(i for i, b in enumerate(blocks) if int(b.window_size) == 0)
            window = max(int(b.window_size) for b in blocks)
            mlp_ratio = int(blocks[0].mlp.lin1.out_features // embed_dim)
            img, patch = int(image_encoder.img_size), int(image_encoder.patch_embed.proj.kernel_size[0])
            out_chans = int(image_encoder.neck[0].out_channels)
        name = {(768, 12): "vit_b", (1024, 24): "vit_l", (1280, 32): "vit_h"}.get((embed_dim, depth), f"vit_{embed_dim}x{depth}")
        extra = {}
        if pixel_mean is not None:
            extra["pixel_mean"] = tuple(float(v) for v in pixel_mean)
        if pixel_std is not None:
            extra["pixel_std"] = tuple(float(v) for v in pixel_std)
        if "image_embedding_size" in kw and int(kw["image_embedding_size"]) != img // patch:
            raise ValueError("image_embedding_size must equal image_size / vit_patch_size")
        return SamConfig(name, embed_dim, depth, heads, gidx, img_size=img, patch_size=patch, window_size=window,
                         mlp_ratio=mlp_ratio, out_chans=out_chans, **extra)

    @property
    def device(self):
        return self.pixel_mean.device


class SamPredictor:
    max_prompt_points = 4000          # SAMPT_DEC_MAX_POINTS (include/sampt_hip.h)

    def __init__(self, sam_model: SamHip):
        self.model = sam_model
        self.transform = ResizeLongestSide(sam_model.cfg.img_size)
        self._vit = self._dec = None
        self._dev = None
        self._ws_vit: Dict[int, torch.Tensor] = {}
        self._ws_dec: Dict[Tuple[int, int], torch.Tensor] = {}
        self._ws_hq = None
        self._stage: Dict[tuple, Dict[str, torch.Tensor]] = {}
        # hipGraph replay of the per-(frame, object) decode chain (sampt_sam_track_decode_graph); SAMPT_DEC_GRAPH=0|1
        self.use_graph = os.environ.get("SAMPT_DEC_GRAPH", "1") != "0"
        # skip the frame-independent padding rows of landscape frames in the blocks before the first global one (exact)
        self.skip_dead_rows = os.environ.get("SAMPT_VIT_SKIP_DEAD", "1") != "0"
        self._dead_cache = {}
        # fp16 ViT mode: static bias correction of the weight rounding, calibrated once per frame geometry (see
        # ``_select_bias_set``); SAMPT_VIT_BIAS_CORR=0 disables
        self.bias_correction = os.environ.get("SAMPT_VIT_BIAS_CORR", "1") != "0"
        self._bias_orig: Optional[Dict[str, torch.Tensor]] = None
        self._bias_sets: Dict[Tuple[int, int], Dict[str, torch.Tensor]] = {}
        self._bias_live = None
        self.reset_image()
        self.stats = {"set_image": 0, "predict": 0, "encoded_frames": 0, "bias_calibrations": 0}

    @property
    def device(self):
        return self.model.device

    def reset_image(self):
        self.is_image_set = False
        self.features = None
        self._feat_tokens = self._hq_tokens = None
        self.original_size = self.input_size = None

    # -- engines -------------------------------------------------------------------------------------
    @_lib.on_device(lambda self, *a, **k: self.model.device)
    def _ensure(self):
        dev = self.model.device
        if self._vit is not None and self._dev == dev:
            return
        _lib.require_hip(dev, "SamPredictor")
        lib = _lib.load()
        if self._vit is not None:      # the model moved to another device: engines, scratch and caches of the old one go
            with _lib.device_guard(self._dev):
                torch.cuda.synchronize(self._dev)
                lib.sampt_vit_destroy(self._vit)
                lib.sampt_dec_destroy(self._dec)
            self._vit = self._dec = None
            self._ws_vit, self._ws_dec, self._ws_hq = {}, {}, None
            self._stage.clear()
            self._dead_cache.clear()
            self._bias_orig, self._bias_live = None, None
            self._bias_sets.clear()
            self.reset_image()
        cfg, m = self.model.cfg, self.model
        f16 = {"f32": 0, "f16": 1, "f16x3": 2}[m.precision]
        self._wv = pack_vit(m.sd, cfg, dev, f16, m.max_batch)
        c = _lib.VitConfigC()
        c.embed_dim, c.depth, c.num_heads, c.grid, c.window = cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.grid, cfg.window_size
        c.patch, c.out_chans, c.mlp_ratio, c.img_size = cfg.patch_size, cfg.out_chans, cfg.mlp_ratio, cfg.img_size
        c.global_mask = sum(1 << i for i in cfg.global_attn_indexes)
        c.f16 = f16
        for i in range(3):
            c.pixel_mean[i], c.pixel_std[i] = cfg.pixel_mean[i], cfg.pixel_std[i]
        names, ptrs, n = _lib.name_table(self._wv)
        h = C.c_void_p()
        _lib.check(lib.sampt_vit_create(C.byref(c), names, ptrs, n, m.max_batch, C.byref(h)), "sampt_vit_create")
        self._vit = h
        self._wd = pack_decoder(m.sd, cfg, dev, m.max_decode_batch, hq=m.hq)
        names, ptrs, n = _lib.name_table(self._wd)
        h2 = C.c_void_p()
        _lib.check(lib.sampt_dec_create(names, ptrs, n, cfg.grid, cfg.img_size, m.max_decode_batch,
                                        cfg.embed_dim if m.hq else 0, C.byref(h2)), "sampt_dec_create")
        self._dec, self._dev, self._lib = h2, dev, lib

    def __del__(self):
        try:
            if self._vit is not None:
                self._lib.sampt_vit_destroy(self._vit)
            if self._dec is not None:
                self._lib.sampt_dec_destroy(self._dec)
        except Exception:
            pass

    @_lib.on_device(lambda self, *a, **k: self.model.device)
    def set_gemm_workgroups(self, per_xcd):
        """Persistent GEMM workgroups per XCD (of 32 CUs) for the following ``encode_frames`` calls; 0 = one per CU.  An int, or
        four ints (qkv, proj, fc1, fc2): one count per launch kind.  See sampt_vit_set_gemm_workgroups(_kind)
        (include/sampt_hip.h)."""
        self._ensure()
        if isinstance(per_xcd, (tuple, list)):
            q, p, f1, f2 = (int(v) for v in per_xcd)
            _lib.check(self._lib.sampt_vit_set_gemm_workgroups_kind(self._vit, q, p, f1, f2), "sampt_vit_set_gemm_workgroups_kind")
            return
        _lib.check(self._lib.sampt_vit_set_gemm_workgroups_kind(self._vit, 0, 0, 0, 0), "sampt_vit_set_gemm_workgroups_kind")
        _lib.check(self._lib.sampt_vit_set_gemm_workgroups(self._vit, int(per_xcd)), "sampt_vit_set_gemm_workgroups")

    @_lib.on_device(lambda self, *a, **k: self.model.device)
    def gemm_profile_begin(self):
        """Start timing every fp16 GEMM launch of the image encoder with HIP events (see sampt_vit_profile_begin)."""
        self._ensure()
        _lib.check(self._lib.sampt_vit_profile_begin(self._vit), "sampt_vit_profile_begin")

    @_lib.on_device(lambda self, *a, **k: self.model.device)
    def gemm_profile_end(self):
        """-> (algorithmic FLOP, kernel milliseconds, launches) since gemm_profile_begin."""
        fl, ms, n = C.c_double(), C.c_double(), C.c_int()
        _lib.check(self._lib.sampt_vit_profile_end(self._vit, C.byref(fl), C.byref(ms), C.byref(n)), "sampt_vit_profile_end")
        return fl.value, ms.value, n.value

    def _select_bias_set(self, H: int, W: int) -> None:
        """Static bias correction of the fp16 mode's WEIGHT rounding ("fp16 error budget").

        Rounding a weight matrix to fp16 adds A.(W - fp16(W))^T to a GEMM's output.  The part of that error that is the same for
        every token — mean_tokens(A).(W - fp16(W))^T, one constant per output column — is the part attention's averaging over
        keys does not damp, and it is 35 % of the mode's embedding error (tools/f16_error_budget.py; measured on the ViT-B bench
        frame: rms 6.0e-4 -> 3.9e-4 with it removed).  The token means hardly depend on the picture — they follow the frame
        GEOMETRY (which token rows are zero padding) and generic image statistics: calibrated on a seeded uniform-noise frame of
        the same size the correction removes as much as the frame's own means do — so it is folded into the four bias vectors
        of every block ONCE per frame geometry at no cost per frame: one calibration pass of the encoder over the noise frame
        records the column means of every GEMM's A operand (sampt_vit_calibrate), the host adds (W - fp16(W)).mean in fp64 to
        the fp32 biases.  The qkv of SAM's zero-padded window tokens keeps the ORIGINAL bias (a zero input has no weight-rounding
        error; the attention kernels read it from the separate fp16 bias row).  Deterministic: the same geometry always gets
        the same biases, whatever was encoded before."""
        m = self.model
        if not self.bias_correction or m.precision != "f16":
            if self._bias_live is not None:       # switched off at run time: the original biases come back, and with them go
                for k, v in self._bias_orig.items():      # the dead rows that were computed under the corrected ones
                    self._wv[k].copy_(v)
                self._bias_live = None
                self._dead_cache.clear()
            return
        cfg, key = m.cfg, (int(H), int(W))
        e = "image_encoder.blocks."
        names = [f"{e}{i}.{mod}.bias" for i in range(cfg.depth) for mod in VIT_GEMM_KINDS]
        if self._bias_orig is None:
            self._bias_orig = {k: self._wv[k].clone() for k in names}
        if key not in self._bias_sets:
            for k in names:                                   # calibrate on the uncorrected weights, whatever ran before
                self._wv[k].copy_(self._bias_orig[k])
            self._bias_live = None
            D, ld = cfg.embed_dim, cfg.mlp_ratio * cfg.embed_dim
            g = torch.Generator().manual_seed(0x5A17)
            frame = torch.randint(0, 256, (1, 3, key[0], key[1]), generator=g, dtype=torch.uint8).to(self._dev)
            cal = torch.zeros((cfg.depth, 4, ld), dtype=torch.float32, device=self._dev)
            out = torch.empty((1, cfg.grid * cfg.grid, cfg.out_chans), dtype=torch.float32, device=self._dev)
            interm = torch.empty((1, cfg.grid * cfg.grid, D), dtype=torch.float32, device=self._dev) if m.hq else None
            ws = self._vit_ws(1)
            _lib.check(self._lib.sampt_vit_calibrate(self._vit, _lib.ptr(cal), ld), "sampt_vit_calibrate")
            try:
                _lib.check(self._lib.sampt_vit_encode(self._vit, _lib.ptr(frame), 1, 1, key[0], key[1], _lib.ptr(out), _lib.ptr(interm),
                                                      _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "sampt_vit_encode(calibration)")
            finally:
                _lib.check(self._lib.sampt_vit_calibrate(self._vit, None, 0), "sampt_vit_calibrate(end)")
            bset = {k: v.to(self._dev) for k, v in vit_bias_correction(m.sd, cfg, cal).items()}    # host, fp64 (pack.py)
            while len(self._bias_sets) >= 8:                   # least recently used geometry goes, not all of them
                self._bias_sets.pop(next(iter(self._bias_sets)))
            self._bias_sets[key] = bset
            self.stats["bias_calibrations"] += 1
        self._bias_sets[key] = self._bias_sets.pop(key)          # most recently used last
        if self._bias_live != key:
            if self._bias_live is None:
                self._dead_cache.pop(key, None)               # rows computed while the correction was switched off
            for k, v in self._bias_sets[key].items():
                self._wv[k].copy_(v)
            self._bias_live = key
            # (the dead-row cache of a geometry is computed after its biases are in place and keyed by the geometry)

    def _vit_ws(self, B: int) -> torch.Tensor:
        if B not in self._ws_vit:
            n = C.c_size_t()
            _lib.check(self._lib.sampt_vit_encode_workspace_bytes(self._vit, B, C.byref(n)), "vit_workspace")
            self._ws_vit = {B: torch.empty(n.value, dtype=torch.uint8, device=self._dev)}  # keep only the latest size
        return self._ws_vit[B]

    def _dead_rows(self, frames: torch.Tensor, chw: bool, H: int, W: int, ws: torch.Tensor) -> Optional[torch.Tensor]:
        """Residual stream of the token rows no pixel of an (H, W) frame reaches (the zero padding of Sam.preprocess below
        a landscape frame) at the input of the first global-attention block: the same in every frame, computed once per
        geometry from the first frame that has it (sampt_vit_encode_live, include/sampt_hip.h).  None: nothing to skip."""
        key = (H, W)      # (an entry is always computed under its geometry's bias set: _select_bias_set drops it on a toggle)
        if key not in self._dead_cache:
            lh, nb = C.c_int(), C.c_size_t()
            _lib.check(self._lib.sampt_vit_live_rows(self._vit, H, W, C.byref(lh), C.byref(nb)), "sampt_vit_live_rows")
            cache = None
            if nb.value:
                cache = torch.empty(nb.value // 4, dtype=torch.float32, device=self._dev)
                _lib.check(self._lib.sampt_vit_encode_live(self._vit, _lib.ptr(frames[:1]), 1 if chw else 0, 1, H, W, None,
                                                           None, _lib.ptr(cache), 1, _lib.ptr(ws), ws.numel(),
                                                           _lib.stream_ptr()), "sampt_vit_encode_live(build)")
            while len(self._dead_cache) >= 8:
                self._dead_cache.pop(next(iter(self._dead_cache)))
            self._dead_cache[key] = cache
        return self._dead_cache[key]

    def _dec_ws(self, oh: int, ow: int, frames: int = 1, k: int = 0) -> torch.Tensor:
        """Decoder scratch for `frames` items with up to k prompt points each (sized in steps: 120 points cover every
        shipped SAM-PT configuration; larger prompts — many objects feeding each other negatives, the VIS adapter's
        mask batches — grow it, the reference accepts any k)."""
        key = (oh, ow)
        k = 120 if k <= 120 else min(-(-k // 256) * 256, self.max_prompt_points)   # never past SAMPT_DEC_MAX_POINTS
        have = self._ws_dec.get(key)
        if have is None or have[0] < frames or have[1] < k:
            if have is not None:
                frames, k = max(frames, have[0]), max(k, have[1])
            n = C.c_size_t()
            _lib.check(self._lib.sampt_dec_workspace_bytes_k(self._dec, frames, k, oh, ow, C.byref(n)), "dec_workspace")
            self._ws_dec = {key: (frames, k, torch.empty(n.value, dtype=torch.uint8, device=self._dev))}
        return self._ws_dec[key][2]

    # -- image encoder -------------------------------------------------------------------------------
    @torch.no_grad()
    @_lib.on_device(lambda self, *a, **k: self.model.device)
    def encode_frames(self, frames: torch.Tensor, chw: bool = True, batch_events: Optional[list] = None,
                      gemm_workgroups=None):
        """frames uint8 (T,3,H,W) [chw] or (T,H,W,3) on device -> token-major embeddings (T, grid*grid, 256) f32; for an
        HQ-SAM model a ``ClipFeatures`` that also carries the per-frame HQ features (T, 16*grid*grid, 32).
        ``batch_events``: a list that receives one ``(end_frame, torch.cuda.Event)`` per encoder batch, recorded on the
        current stream when frames [.., end_frame) are done — lets a caller start decoding early frames on another stream
        while later batches are still being encoded.  ``gemm_workgroups``: persistent GEMM workgroups per XCD for the encoder
        batches of this call — an int, or a sequence with one entry per batch (the last one repeats); see
        ``set_gemm_workgroups``.  Reset to the default (one per CU) afterwards."""
        self._ensure()
        frames = frames.to(self._dev).contiguous()
        frames = self.transform.apply_image_torch(frames, chw=chw)      # PIL-exact resize when the longest side != img_size
        T = frames.shape[0]
        H, W = (frames.shape[2], frames.shape[3]) if chw else (frames.shape[1], frames.shape[2])
        g, Cc = self.model.cfg.grid, self.model.cfg.out_chans
        out = torch.empty((T, g * g, Cc), dtype=torch.float32, device=self._dev)
        Bm = min(self.model.max_batch, self.model.max_decode_batch)
        hq = interm = None
        if self.model.hq:
            hq = torch.empty((T, 16 * g * g, Cc // 8), dtype=torch.float32, device=self._dev)
            interm = torch.empty((min(Bm, T), g * g, self.model.cfg.embed_dim), dtype=torch.float32, device=self._dev)
        self._select_bias_set(H, W)
        dead = self._dead_rows(frames, chw, H, W, self._vit_ws(min(Bm, T))) if self.skip_dead_rows else None
        # one entry per batch (the last repeats); an entry is an int or a (qkv, proj, fc1, fc2) tuple
        wgs = None if gemm_workgroups is None else ([int(gemm_workgroups)] if isinstance(gemm_workgroups, int)
                                                      else [tuple(int(x) for x in v) if isinstance(v, (tuple, list)) else int(v)
                                                            for v in gemm_workgroups])
        for bi, t0 in enumerate(range(0, T, Bm)):
            B = min(Bm, T - t0)
            ws = self._vit_ws(B)
            if wgs:
                self.set_gemm_workgroups(wgs[min(bi, len(wgs) - 1)])
            if dead is not None:
                _lib.check(self._lib.sampt_vit_encode_live(self._vit, _lib.ptr(frames[t0:t0 + B]), 1 if chw else 0, B, H, W,
                                                           _lib.ptr(out[t0:t0 + B]), _lib.ptr(interm), _lib.ptr(dead), 0,
                                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                           "sampt_vit_encode_live")
            else:
                _lib.check(self._lib.sampt_vit_encode(self._vit, _lib.ptr(frames[t0:t0 + B]), 1 if chw else 0, B, H, W,
                                                      _lib.ptr(out[t0:t0 + B]), _lib.ptr(interm), _lib.ptr(ws), ws.numel(),
                                                      _lib.stream_ptr()), "sampt_vit_encode")
            if hq is not None:      # the ViT tap is consumed batch by batch: only the 32-channel HQ features stay resident
                if self._ws_hq is None or self._ws_hq[0] < B:
                    n = C.c_size_t()
                    _lib.check(self._lib.sampt_dec_hq_workspace_bytes(self._dec, B, C.byref(n)), "hq_workspace")
                    self._ws_hq = (B, torch.empty(n.value, dtype=torch.uint8, device=self._dev))
                wsh = self._ws_hq[1]
                _lib.check(self._lib.sampt_dec_hq_features(self._dec, B, _lib.ptr(out[t0:t0 + B]), _lib.ptr(interm),
                                                           _lib.ptr(hq[t0:t0 + B]), _lib.ptr(wsh), wsh.numel(),
                                                           _lib.stream_ptr()), "sampt_dec_hq_features")
            if batch_events is not None:
                ev = torch.cuda.Event()
                ev.record()
                batch_events.append((t0 + B, ev))
        if wgs:
            self.set_gemm_workgroups(0)
        self.stats["encoded_frames"] += T
        return ClipFeatures(out, hq) if hq is not None else out

    def set_features(self, feat_tokens, original_size, input_size=None):
        """Install a pre-computed embedding (one frame of ``encode_frames``) as the current image."""
        g, Cc = self.model.cfg.grid, self.model.cfg.out_chans
        if isinstance(feat_tokens, ClipFeatures):
            feat_tokens, self._hq_tokens = feat_tokens.emb, feat_tokens.hq.contiguous()
        elif self.model.hq:
            raise ValueError("HQ-SAM: set_features needs the ClipFeatures item returned by encode_frames")
        self._feat_tokens = feat_tokens.contiguous()
        self.features = self._feat_tokens.view(g, g, Cc).permute(2, 0, 1).unsqueeze(0)  # (1,256,g,g) view, as upstream
        self.original_size = tuple(original_size)
        self.input_size = tuple(input_size) if input_size is not None else tuple(original_size)
        self.is_image_set = True

    @torch.no_grad()
    @_lib.on_device(lambda self, *a, **k: self.model.device)
    def set_image(self, image: np.ndarray, image_format: str = "RGB") -> None:
        assert image_format in ("RGB", "BGR")
        if image_format != self.model.image_format:
            image = image[..., ::-1]
        H, W = image.shape[:2]
        self._ensure()
        t = torch.as_tensor(np.ascontiguousarray(image), device=self._dev)
        self.stats["set_image"] += 1
        from . import prefetch
        item = prefetch.lookup(self, t)          # a frame of the clip the tracker was given: batch-encoded once (prefetch.py)
        if item is not None:
            self.set_features(item, (H, W), self.transform.get_preprocess_shape(H, W, self.transform.target_length))
            return
        t = self.transform.apply_image_torch(t)                      # identity when the longest side is already img_size
        feats = self.encode_frames(t[None], chw=False)
        self.set_features(feats[0], (H, W), tuple(t.shape[:2]))      # (ClipFeatures[0] for HQ-SAM)

    # -- prompt encoder + mask decoder ---------------------------------------------------------------
    @torch.no_grad()
    @_lib.on_device(lambda self, *a, **k: self.model.device)
    def predict_torch(self, point_coords, point_labels, boxes=None, mask_input=None, multimask_output: bool = True,
                      return_logits: bool = False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        if multimask_output and self.model.hq:
            raise NotImplementedError("multimask_output=True with the HQ-SAM decoder is not built (SAM-PT never asks for "
                                      "it: sam_pt.py:787, 796, 805, 826)")
        if point_coords is None:
            raise NotImplementedError("predict_torch: prompts without points are not supported")
        self._ensure()
        dev = self._dev
        oh, ow = self.original_size
        ih, iw = self.input_size
        B = point_coords.shape[0]                      # prompts against the current image (upstream's batch dimension)
        pts = point_coords.to(dev, torch.float32).contiguous()
        lab = point_labels.to(dev, torch.int32).contiguous()
        box = boxes.reshape(B, 4).to(dev, torch.float32).contiguous() if boxes is not None else None
        L = 4 * self.model.cfg.grid
        mi = mask_input.reshape(B, L, L).to(dev, torch.float32).contiguous() if mask_input is not None else None
        nm = 3 if multimask_output else 1
        logits = torch.empty((B, nm, oh, ow), dtype=torch.float32, device=dev)
        iou = torch.empty((B, nm), dtype=torch.float32, device=dev)
        low = torch.empty((B, nm, L, L), dtype=torch.float32, device=dev)
        k = pts.shape[1]
        ws = self._dec_ws(oh, ow, 1, k)
        for b in range(B):                             # stream-ordered, no host synchronisation between prompts
            # every prompt gets its own (allocator-aligned) buffers, so a batch member sees exactly what a single call sees
            one = B == 1
            p_b, l_b = (pts[0], lab[0]) if one else (pts[b].clone(), lab[b].clone())
            b_b = None if box is None else (box[0] if one else box[b].clone())
            m_b = None if mi is None else (mi[0] if one else mi[b].clone())
            lg, io, lw = (logits[0], iou[0], low[0]) if one else (torch.empty_like(logits[0]), torch.empty_like(iou[0]),
                                                                  torch.empty_like(low[0]))
            if multimask_output:
                _lib.check(self._lib.sampt_sam_decode_multimask(self._dec, _lib.ptr(self._feat_tokens), _lib.ptr(p_b),
                                                                _lib.ptr(l_b), k, _lib.ptr(b_b), _lib.ptr(m_b), ih, iw, oh, ow,
                                                                _lib.ptr(lg), _lib.ptr(io), _lib.ptr(lw),
                                                                _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                           "sampt_sam_decode_multimask")
            else:
                _lib.check(self._lib.sampt_sam_decode(self._dec, _lib.ptr(self._feat_tokens), _lib.ptr(self._hq_tokens),
                                                      _lib.ptr(p_b), _lib.ptr(l_b), k, _lib.ptr(b_b), _lib.ptr(m_b), ih, iw,
                                                      oh, ow, _lib.ptr(lg), _lib.ptr(io), _lib.ptr(lw), _lib.ptr(ws),
                                                      ws.numel(), _lib.stream_ptr()), "sampt_sam_decode")
            if not one:
                logits[b].copy_(lg), iou[b].copy_(io), low[b].copy_(lw)
            self.stats["predict"] += 1
        masks = logits if return_logits else logits > self.model.mask_threshold
        return masks, iou, low

    def predict(self, point_coords=None, point_labels=None, box=None, mask_input=None, multimask_output=True,
                return_logits=False):
        """numpy flavour (App. A-1; used at sam_pt/vos_eval/eval.py:244-250)."""
        dev = self.model.device
        pc = torch.as_tensor(self.transform.apply_coords(point_coords, self.original_size), dtype=torch.float, device=dev)[None]
        pl = torch.as_tensor(point_labels, dtype=torch.int, device=dev)[None]
        bx = None
        if box is not None:
            bx = torch.as_tensor(self.transform.apply_boxes(box, self.original_size), dtype=torch.float, device=dev)[None]
        mi = torch.as_tensor(mask_input, dtype=torch.float, device=dev)[None] if mask_input is not None else None
        m, i, l = self.predict_torch(pc, pl, bx, mi, multimask_output, return_logits)
        return m[0].cpu().numpy(), i[0].cpu().numpy(), l[0].cpu().numpy()

    def decode_staging(self, F: int, ld_pts: int, size_hw) -> Dict[str, torch.Tensor]:
        """Persistent input / output buffers of one prompt bucket (F items, ld_pts point slots, frame size): a captured
        hipGraph replays fixed pointers, so the decode chain of a bucket always reads and writes these tensors."""
        self._ensure()
        key = (F, ld_pts, tuple(size_hw))
        st = self._stage.pop(key, None)                    # (re-inserted below: dict order = least recently used first)
        if st is None:
            g, Cc, dev = self.model.cfg.grid, self.model.cfg.out_chans, self._dev
            # the small per-item inputs are ONE device block [pts | labels | k_item | npos_item | item numbers] mirrored by a
            # pinned host block: the host packs a chunk's prompts straight into the mirror and one asynchronous copy moves them
            # (SamPt._enqueue_sam_fused) — no device-side gathers, one H2D per chunk
            words = F * ld_pts * 2 + F * ld_pts + 3 * F
            block = torch.empty((words,), dtype=torch.int32, device=dev)
            o0, o1, o2, o3, o4 = 0, F * ld_pts * 2, F * ld_pts * 3, F * ld_pts * 3 + F, F * ld_pts * 3 + 2 * F
            offs = (o0, o1, o2, o3, o4)

            def mirror(k, _m=[], _offs=offs, _F=F, _ld=ld_pts, _words=words, _cuda=dev.type == "cuda"):
                """Pinned host mirror number k of the block (a clip with several chunks in this bucket packs each into its own: the
                device block is re-used in stream order, a host buffer must stay untouched until its copy has run)."""
                while len(_m) <= k:
                    h = torch.empty((_words,), dtype=torch.int32, pin_memory=True) if _cuda else torch.empty((_words,), dtype=torch.int32)
                    hn = h.numpy()
                    a0, a1, a2, a3, a4 = _offs
                    _m.append({"host": h, "free": None, "pts": hn[a0:a1].view(np.float32).reshape(_F, _ld, 2),
                               "labels": hn[a1:a2].reshape(_F, _ld), "k": hn[a2:a3], "npos": hn[a3:a4], "items": hn[a4:]})
                return _m[k]

            st = {"feats": torch.empty((F, g * g, Cc), dtype=torch.float32, device=dev),
                  "block": block, "mirror": mirror,
                  "pts": block[o0:o1].view(torch.float32).view(F, ld_pts, 2),
                  "labels": block[o1:o2].view(F, ld_pts),
                  "k_item": block[o2:o3], "npos_item": block[o3:o4], "items": block[o4:],
                  "logits": torch.empty((F,) + tuple(size_hw), dtype=torch.float32, device=dev),
                  "score": torch.empty((F,), dtype=torch.float32, device=dev)}
            if self.model.hq:
                st["hq"] = torch.empty((F, 16 * g * g, Cc // 8), dtype=torch.float32, device=dev)
        self._stage[key] = st
        # a handful of buckets is all a clip needs; beyond 8 buckets or SAMPT_STAGE_MAX_BYTES (default 4 GiB: one F = 128
        # bucket of HQ-SAM at 480p is 1.7 GB) the least recently used ones go (a bucket still being read by an enqueued chain
        # is safe to drop: its memory returns to the allocator of the stream that chain runs on)
        cap = int(os.environ.get("SAMPT_STAGE_MAX_BYTES", str(4 << 30)))
        while len(self._stage) > 1 and (len(self._stage) > 8 or self._stage_bytes() > cap):
            self._stage.pop(next(iter(self._stage)))
        return st

    def _stage_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for st in self._stage.values() for k, t in st.items()
                   if k in ("feats", "block", "logits", "score", "hq"))

    def graph_stats(self):
        """(cached graphs, captures, replays) of the decoder handle."""
        a, b, c = C.c_long(), C.c_long(), C.c_long()
        _lib.check(self._lib.sampt_dec_graph_stats(self._dec, C.byref(a), C.byref(b), C.byref(c)), "sampt_dec_graph_stats")
        return a.value, b.value, c.value

    @torch.no_grad()
    @_lib.on_device(lambda self, *a, **k: self.model.device)
    def track_decode(self, feat_tokens: torch.Tensor, pts: torch.Tensor, labels: torch.Tensor, k: int, n_pos_first: int,
                     refine_iters: int, iou_thr: float, size_hw, out_logits: torch.Tensor, out_score: torch.Tensor,
                     k_item: Optional[torch.Tensor] = None, npos_item: Optional[torch.Tensor] = None, graph: bool = False):
        """SamPt.predict_mask (sam_pt.py:760-837) for F independent (frame, object) items that share the visible-point
        count k, as ONE batched device-side chain without host syncs.  feat_tokens (F,g*g,256); pts (F,ld,2) f32 in
        input-frame px and labels (F,ld) i32 with the first k entries valid (positives first); n_pos_first = -1 for
        the single-pass case, else the number of leading positives used by the positives-only first pass.
        Ragged batches: ``k_item`` / ``npos_item`` (F,) int32 device tensors give every item its own point / leading
        positive count (<= k / n_pos_first); padding tokens are masked on the device, results equal the un-batched ones.
        Results are written into out_logits (F,H,W) and out_score (F,)."""
        self._ensure()
        oh, ow = size_hw
        ih, iw = self.transform.get_preprocess_shape(oh, ow, self.model.cfg.img_size)   # size the frames were encoded at
        hq_tokens = None
        if isinstance(feat_tokens, ClipFeatures):
            feat_tokens, hq_tokens = feat_tokens.emb, feat_tokens.hq
        F = feat_tokens.shape[0]
        assert F <= self.model.max_decode_batch
        ws = self._dec_ws(oh, ow, F, k)
        fn = self._lib.sampt_sam_track_decode_graph if graph else self._lib.sampt_sam_track_decode
        _lib.check(fn(self._dec, F, _lib.ptr(feat_tokens), _lib.ptr(hq_tokens),
                                                    _lib.ptr(pts), _lib.ptr(labels),
                                                    k, _lib.ptr(k_item), _lib.ptr(npos_item), pts.shape[1], n_pos_first,
                                                    refine_iters, float(iou_thr), ih, iw,
                                                    oh, ow, _lib.ptr(out_logits), _lib.ptr(out_score), _lib.ptr(ws),
                                                    ws.numel(), _lib.stream_ptr()), "sampt_sam_track_decode")
