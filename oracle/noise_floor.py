"""How far the ORACLE moves under a perturbation of its own weights that no implementation could resolve.
TEST INFRASTRUCTURE ONLY (see oracle/parity.py).

A tracker with random weights that chains many windows (CoTracker: a window starts from the previous window's estimate,
cotracker/tracker.py:72-104) can amplify fp32 round-off until "identical after round()" is not a property any implementation
can have.  The conditioned parity workloads (oracle/workloads.py) avoid that regime; this module MEASURES it instead, so that a
HIP-vs-oracle distance on an unconditioned workload can be read against the oracle's own noise floor:

    run A : the oracle on the seeded weights
    run B : the same oracle on weights multiplied by (1 + rel * N(0, 1)), rel = 1e-7 (below fp32's 6e-8 unit round-off per
            operation once it has passed through a handful of layers)

``tracker_noise_floor`` returns both runs' trajectories / visibilities and their distance.  A device result whose distance to
run A is of the size of |A - B| is as close to the oracle as the oracle is to itself.
"""
from __future__ import annotations

from typing import Callable, Dict

import torch


def perturbed(sd: Dict[str, torch.Tensor], rel: float = 1e-7, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Every floating-point tensor of ``sd`` times (1 + rel * N(0, 1)), seeded."""
    g = torch.Generator().manual_seed(seed)
    out = type(sd)()
    for k, v in sd.items():
        if v.is_floating_point():
            out[k] = (v.double() * (1.0 + rel * torch.randn(v.shape, generator=g, dtype=torch.float64))).to(v.dtype)
        else:
            out[k] = v.clone()
    return out


def distance(tr_a, vi_a, tr_b, vi_b) -> Dict:
    """max |dtraj| in px, coordinates whose round() differs, visibilities that differ."""
    tr_a, tr_b = tr_a.cpu().float(), tr_b.cpu().float()
    return {"traj_max_abs_px": float((tr_a - tr_b).abs().max()),
            "traj_index_differing": int((tr_a.round() != tr_b.round()).sum()),
            "coords": int(tr_a.numel()),
            "vis_differing": int((vi_a.cpu().bool() != vi_b.cpu().bool()).sum())}


def tracker_noise_floor(make_tracker: Callable[[Dict], object], sd: Dict[str, torch.Tensor], rgbs: torch.Tensor,
                        queries: torch.Tensor, rel: float = 1e-7) -> Dict:
    """``make_tracker(state_dict)`` -> an oracle tracker with ``forward(rgbs (1,T,3,H,W), queries (1,N,3))``.
    -> {"a": (traj, vis), "b": (traj, vis), "floor": distance(a, b), "rel": rel}"""
    with torch.no_grad():
        tr_a, vi_a = make_tracker(sd).forward(rgbs, queries)
        tr_b, vi_b = make_tracker(perturbed(sd, rel)).forward(rgbs, queries)
    return {"a": (tr_a, vi_a), "b": (tr_b, vi_b), "floor": distance(tr_a, vi_a, tr_b, vi_b), "rel": rel}
