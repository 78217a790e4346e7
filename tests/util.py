"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes as C

import numpy as np
import torch


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def max_abs(a, b) -> float:
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def iou(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.cpu().bool(), b.cpu().bool()
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u


from sam_pt_amd.synth import disc_queries, synthetic_clip  # noqa: E402,F401
