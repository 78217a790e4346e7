"""ctypes binding of libsampt_hip.so (C ABI: include/sampt_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import functools
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsampt_hip.so")

c_void_p, c_int, c_size_t, c_float, c_char_p = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_char_p


class SamptError(RuntimeError):
    pass


class VitConfigC(C.Structure):
    _fields_ = [("embed_dim", c_int), ("depth", c_int), ("num_heads", c_int), ("grid", c_int), ("window", c_int),
                ("patch", c_int), ("out_chans", c_int), ("mlp_ratio", c_int), ("img_size", c_int),
                ("global_mask", c_int), ("f16", c_int), ("pixel_mean", c_float * 3), ("pixel_std", c_float * 3)]


# name -> (restype, argtypes); every symbol declared in include/sampt_hip.h
_P = c_void_p
_SIGS = {
    "sampt_version": (c_int, []),
    "sampt_last_error": (c_char_p, []),
    "sampt_pips_create": (c_int, [C.POINTER(c_char_p), C.POINTER(_P), c_int, c_int, c_int, C.POINTER(_P)]),
    "sampt_pips_destroy": (None, [_P]),
    "sampt_pips_fnet_workspace_bytes": (c_int, [_P, c_int, c_int, c_int, C.POINTER(c_size_t)]),
    "sampt_pips_fnet_f32": (c_int, [_P, _P, c_int, c_int, c_int, C.POINTER(_P), _P, c_size_t, _P]),
    "sampt_pips_sample_feat_f32": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, _P]),
    "sampt_pips2_create": (c_int, [C.POINTER(c_char_p), C.POINTER(_P), c_int, c_int, C.POINTER(_P)]),
    "sampt_pips2_destroy": (None, [_P]),
    "sampt_pips2_fnet_workspace_bytes": (c_int, [_P, c_int, c_int, c_int, C.POINTER(c_size_t)]),
    "sampt_pips2_fnet_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, C.POINTER(_P), _P, c_size_t, _P]),
    "sampt_pips2_update_workspace_bytes": (c_int, [_P, c_int, c_int, C.POINTER(c_size_t)]),
    "sampt_pips2_update_f32": (c_int, [_P, C.POINTER(_P), c_int, c_int, _P, c_int, c_int, _P, c_int, C.POINTER(_P), c_int, _P,
                                       _P, c_size_t, _P]),
    "sampt_cotracker_create": (c_int, [C.POINTER(c_char_p), C.POINTER(_P), c_int, c_int, c_int, C.POINTER(_P)]),
    "sampt_cotracker_destroy": (None, [_P]),
    "sampt_resize_frames_f32": (c_int, [_P, c_int, C.c_long, c_int, c_int, _P, c_int, c_int, _P]),
    "sampt_cotracker_fnet_workspace_bytes": (c_int, [_P, c_int, c_int, c_int, C.POINTER(c_size_t)]),
    "sampt_cotracker_fnet_f32": (c_int, [_P, _P, c_int, c_int, c_int, C.POINTER(_P), _P, c_size_t, _P]),
    "sampt_cotracker_track_workspace_bytes": (c_int, [_P, c_int, C.POINTER(c_size_t)]),
    "sampt_cotracker_track_f32": (c_int, [_P, C.POINTER(_P), c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P, c_int, _P, _P,
                                          _P, c_size_t, _P]),
    "sampt_pips_update_workspace_bytes": (c_int, [_P, c_int, C.POINTER(c_size_t)]),
    "sampt_pips_update_f32": (c_int, [_P, C.POINTER(_P), c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P, _P, c_size_t, _P]),
    "sampt_vit_create": (c_int, [C.POINTER(VitConfigC), C.POINTER(c_char_p), C.POINTER(_P), c_int, c_int, C.POINTER(_P)]),
    "sampt_vit_destroy": (None, [_P]),
    "sampt_vit_encode_workspace_bytes": (c_int, [_P, c_int, C.POINTER(c_size_t)]),
    "sampt_vit_profile_begin": (c_int, [_P]),
    "sampt_vit_profile_end": (c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_int)]),
    "sampt_vit_encode": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "sampt_attention_t2i_workspace_bytes": (c_int, [c_int, c_int, c_int, C.POINTER(c_size_t)]),
    "sampt_attention_t2i_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "sampt_vit_live_rows": (c_int, [_P, c_int, c_int, C.POINTER(c_int), C.POINTER(c_size_t)]),
    "sampt_vit_encode_live": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "sampt_dec_create": (c_int, [C.POINTER(c_char_p), C.POINTER(_P), c_int, c_int, c_int, c_int, c_int, C.POINTER(_P)]),
    "sampt_dec_hq_workspace_bytes": (c_int, [_P, c_int, C.POINTER(c_size_t)]),
    "sampt_dec_hq_features": (c_int, [_P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "sampt_dec_destroy": (None, [_P]),
    "sampt_dec_workspace_bytes": (c_int, [_P, c_int, c_int, c_int, C.POINTER(c_size_t)]),
    "sampt_dec_workspace_bytes_k": (c_int, [_P, c_int, c_int, c_int, c_int, C.POINTER(c_size_t)]),
    "sampt_sam_decode": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t,
                                 _P]),
    "sampt_sam_decode_multimask": (c_int, [_P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "sampt_sam_track_decode": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_float, c_int, c_int, c_int,
                                       c_int, _P, _P, _P, c_size_t, _P]),
    "sampt_sam_track_decode_graph": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_float, c_int, c_int,
                                             c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "sampt_dec_graph_stats": (c_int, [_P, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "sampt_postprocess_masks": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "sampt_bbox_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sampt_bbox_from_logits": (c_int, [_P, c_int, c_int, _P, _P, c_size_t, _P]),
    "sampt_resize_logits": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "sampt_pil_resample_u8": (c_int, [_P, _P, C.c_long, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "sampt_index_masks": (c_int, [_P, c_int, C.c_long, _P, _P]),
    "sampt_vos_index_masks": (c_int, [_P, c_int, c_int, C.c_long, _P, _P, _P, _P]),
    "sampt_vos_index_masks_resized": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, _P]),
    "sampt_gemm": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "sampt_pips_track_workspace_bytes": (c_int, [_P, c_int, C.POINTER(c_size_t)]),
    "sampt_pips_track_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_int, _P, _P, _P, c_int,
                                     _P, _P, _P, c_size_t, _P, C.POINTER(c_int)]),
    "sampt_vit_set_gemm_workgroups": (c_int, [_P, c_int]),
    "sampt_vit_set_gemm_workgroups_kind": (c_int, [_P, c_int, c_int, c_int, c_int]),
    "sampt_vit_calibrate": (c_int, [_P, _P, c_int]),
    "sampt_gemm_set_stagger": (c_int, [c_int]),
    "sampt_gemm_set_schedule": (c_int, [c_int]),
    "sampt_gemm_set_thin_min_wgs": (c_int, [c_int]),
    "sampt_conv_set_halo": (c_int, [c_int]),
    "sampt_gemm_set_wres": (c_int, [c_int]),
    "sampt_gemm_set_trim": (c_int, [c_int]),
    "sampt_move_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int, c_int, c_int, c_void_p]),
    "sampt_fill_f32": (c_int, [c_void_p, c_size_t, c_float, c_void_p]),
    "sampt_conv_stem7x7": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "sampt_conv3x3_planes_instnorm_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "sampt_gemm_x3_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sampt_gemm_x3_rows_epi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "sampt_sam_mask_dot": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sampt_pips_set_mixer": (c_int, [c_int, c_int]),
    "sampt_stream_create_cu_range": (c_int, [c_int, c_int, C.POINTER(_P)]),
    "sampt_stream_destroy": (c_int, [_P]),
    "sampt_pips_round_launches": (c_int, [_P]),
    "sampt_pips_mix_mlp_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "sampt_pips_mix_xop_halves": (c_size_t, [c_int]),
    "sampt_pips_mix_pre_f32": (c_int, [_P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sampt_pips_mix_mlp_x3": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "sampt_pips_mix_reduce_f32": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sampt_gemm_ex": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, c_int, c_int, _P]),
    "sampt_conv2d_nhwc": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "sampt_instance_norm_nhwc": (c_int, [_P, c_int, c_int, c_int, c_float, c_int, _P, _P, c_size_t, _P]),
    "sampt_instance_norm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sampt_layernorm": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_int, c_int, _P]),
    "sampt_resize_bilinear_nhwc": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "sampt_avgpool2x2_nhwc": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "sampt_corr_sample_f32": (c_int, [C.POINTER(_P), c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P]),
    "sampt_attention_f32": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "sampt_cotracker_attention_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "sampt_vit_attention_f16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "sampt_vit_attention_x3": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "sampt_split_rows_x3": (c_int, [_P, _P, c_int, c_int, _P]),
    "sampt_vit_window_attention": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "sampt_kmedoids_rowsums_f64": (c_int, [_P, c_int, _P, _P]),
    "sampt_kmedoids_alternate": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P]),
    "sampt_qp_corners_workspace_bytes": (c_int, [c_int, c_int, C.POINTER(c_size_t)]),
    "sampt_qp_erode_u8": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "sampt_qp_shi_tomasi": (c_int, [_P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
}

_lib = None


def exported_symbols() -> List[str]:
    return list(_SIGS)


def load():
    """Load the HIP library; raises SamptError if it has not been built (`python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SamptError(f"{LIB_PATH} not found: build it with `make -C sam_pt_amd/csrc` "
                         "(there is no CPU or PyTorch fallback for the SAM-PT hot path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("SAMPT_GEMM_SCHED"):            # A / B switch of the 8-phase GEMM's stage schedule (sampt_gemm_set_schedule)
        lib.sampt_gemm_set_schedule(int(os.environ["SAMPT_GEMM_SCHED"]))
    if os.environ.get("SAMPT_PIPS_MIXER") or os.environ.get("SAMPT_PIPS_MIXER_WGS") or os.environ.get("SAMPT_PIPS_MIXER_DIAG"):   # csrc/pips_mixer.hip (A / B runs)
        lib.sampt_pips_set_mixer(int(os.environ.get("SAMPT_PIPS_MIXER", "2")), int(os.environ.get("SAMPT_PIPS_MIXER_WGS", "16")))
    if os.environ.get("SAMPT_GEMM_TRIM"):              # A / B switch of the persistent GEMM's workgroup trimming (sampt_gemm_set_trim)
        lib.sampt_gemm_set_trim(int(os.environ["SAMPT_GEMM_TRIM"]))
    if os.environ.get("SAMPT_GEMM_WRES"):              # A / B switch of csrc/gemm_x3_wres.hip (sampt_gemm_set_wres)
        lib.sampt_gemm_set_wres(int(os.environ["SAMPT_GEMM_WRES"]))
    if os.environ.get("SAMPT_CONV_HALO"):              # A / B switch of csrc/conv_halo_x3.hip (sampt_conv_set_halo)
        lib.sampt_conv_set_halo(int(os.environ["SAMPT_CONV_HALO"]))
    if os.environ.get("SAMPT_THIN_MIN_WGS"):          # thin f32 GEMM: tile growth threshold (sampt_gemm_set_thin_min_wgs)
        lib.sampt_gemm_set_thin_min_wgs(int(os.environ["SAMPT_THIN_MIN_WGS"]))
    if os.environ.get("SAMPT_GEMM_STAGGER"):          # experiment knob of the 8-phase GEMM (sampt_gemm_set_stagger)
        lib.sampt_gemm_set_stagger(int(os.environ["SAMPT_GEMM_STAGGER"]))
    _lib = lib
    return lib


def require_hip(device, who: str):
    """The hot path has no CPU / PyTorch fallback: anything but a HIP device is an error."""
    if getattr(device, "type", None) != "cuda":
        raise SamptError(f"{who} runs on the HIP device only (no CPU fallback); got {device}")


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sampt_last_error()
        raise SamptError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """torch's current stream ON `device` (default: the current device — inside ``device_guard`` that is the guarded one)."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def cu_range_stream(device, cu_lo: int, cu_hi: int) -> "torch.cuda.Stream":
    """A torch stream on `device` confined to CUs [cu_lo, cu_hi) of every XCD (sampt_stream_create_cu_range).  The HIP stream lives
    as long as the process (a handful per model; the runtime frees them at exit)."""
    h = c_void_p()
    with device_guard(device):
        check(load().sampt_stream_create_cu_range(int(cu_lo), int(cu_hi), C.byref(h)), "sampt_stream_create_cu_range")
    return torch.cuda.ExternalStream(h.value, device=device)


def device_guard(device):
    """The library launches on the HIP *current* device and never calls hipSetDevice itself, and ``stream_ptr()`` is
    torch's current stream of the current device: every public entry point of the package runs inside this guard so that a
    model living on cuda:N works whatever the caller's current device is (``SamHip().to('cuda:1')`` without
    ``set_device``, gloo-initialised multi-GPU processes)."""
    if device is not None and getattr(device, "type", None) == "cuda":
        return torch.cuda.device(device)
    return contextlib.nullcontext()


def on_device(get_device):
    """Decorator form of ``device_guard``: ``get_device(self, *args, **kwargs)`` names the device of the call."""
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(self, *args, **kwargs):
            with device_guard(get_device(self, *args, **kwargs)):
                return fn(self, *args, **kwargs)
        return wrapper
    return deco


def name_table(named: Dict[str, torch.Tensor]):
    """(names array, ptrs array, n) for the *_create functions; keeps the byte strings alive."""
    keys = list(named)
    names = (c_char_p * len(keys))(*[k.encode() for k in keys])
    ptrs = (_P * len(keys))(*[named[k].data_ptr() for k in keys])
    return names, ptrs, len(keys)


def ptr_array(ts: Sequence[torch.Tensor]):
    return (_P * len(ts))(*[t.data_ptr() for t in ts])
