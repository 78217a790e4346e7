"""Query-point selection from a mask (reference: sam_pt/utils/query_points.py).

Needed by ``SamPt`` in ``query_masks`` mode (every VOS run, sam_pt.py:171-177) and by point re-initialisation
(sam_pt.py:529-533).  Like the reference this is host-side work done once per object per (re)initialisation on at most
1800 sub-sampled mask pixels — not a hot kernel.

* ``extract_random_mask_points`` — same RNG consumption as the reference (one ``torch.randperm`` on the global
  generator, query_points.py:55), so results are bit-identical for the same seed (pinned in tests).
* ``extract_kmedoid_points`` — the reference calls ``sklearn_extra.cluster.KMedoids(n_clusters=N).fit`` (defaults:
  euclidean metric, ``method='alternate'``, ``init='heuristic'``, ``max_iter=300``).  scikit-learn-extra is a
  third-party dependency that is absent here, so ``kmedoids_alternate`` restates its published algorithm; **parity
  unpinned** for this function.
* ``extract_mixed_points`` — the reference's n/4 k-medoid + n/3 Shi-Tomasi + rest random split; with the shipped default
  of one negative point it is a single random point.  Shi-Tomasi corners themselves (``cv2.goodFeaturesToTrack`` on an
  eroded mask) need cv2, which is absent: ``NotImplementedError`` once a Shi-Tomasi share is requested (n >= 3).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch


def extract_random_mask_points(mask: torch.Tensor, n_points_to_select: int) -> torch.Tensor:
    """(H,W) {0,1} mask -> (n,2) float32 (x, y) points drawn uniformly from the mask (query_points.py:29-59)."""
    if mask.sum() == 0:
        print("Warning: mask.sum() == 0 in extract_random_mask_points")
        return torch.zeros((n_points_to_select, 2))
    px = mask.nonzero().float()
    if len(px) < n_points_to_select:
        sel = px.repeat(n_points_to_select // len(px) + 1, 1)[:n_points_to_select]
    else:
        sel = px[torch.randperm(len(px))[:n_points_to_select]]
    return sel.flip(1)


def kmedoids_alternate(X: np.ndarray, n_clusters: int, max_iter: int = 300) -> np.ndarray:
    """Medoid indices of KMedoids(method='alternate', init='heuristic', metric='euclidean')."""
    X = np.asarray(X, dtype=np.float64)
    D = np.sqrt(np.maximum(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1), 0.0))
    medoids = np.argpartition(D.sum(axis=1), n_clusters - 1)[:n_clusters]       # heuristic init: most central points
    for _ in range(max_iter):
        old = medoids.copy()
        labels = np.argmin(D[medoids, :], axis=0)
        for k in range(n_clusters):
            members = np.where(labels == k)[0]
            if len(members) == 0:
                continue
            costs = D[np.ix_(members, members)].sum(axis=1)
            best = int(np.argmin(costs))
            cur = costs[int(np.argmax(members == medoids[k]))]
            if costs[best] < cur:
                medoids[k] = members[best]
        if np.all(old == medoids):
            break
    return medoids


def extract_kmedoid_points(mask: torch.Tensor, n_points_to_select: int, subsample_size: int = 1800) -> torch.Tensor:
    """K-medoid centres of (a random 1800-pixel subsample of) the mask, as (x, y) (query_points.py:62-99)."""
    if mask.sum() == 0:
        print("Warning: mask.sum() == 0 in extract_kmedoid_points")
        return torch.zeros((n_points_to_select, 2))
    px = mask.nonzero().float()
    if len(px) < n_points_to_select:
        sel = px.repeat(n_points_to_select // len(px) + 1, 1)[:n_points_to_select]
    else:
        px = px[torch.randperm(len(px))[:subsample_size]]
        idx = kmedoids_alternate(px.numpy(), n_points_to_select)
        sel = px[torch.as_tensor(idx, dtype=torch.long)].type(torch.float32)
    return sel.flip(1)


def extract_corner_points(image, mask, n_points_to_select):
    """Shi-Tomasi corners inside the (eroded) mask (query_points.py:102-162) — needs cv2.goodFeaturesToTrack / cv2.erode,
    which are absent here and have nothing to be pinned against."""
    raise NotImplementedError("point selection method 'shi-tomasi' needs cv2.goodFeaturesToTrack (absent in this build)")


def extract_mixed_points(query_masks, query_points_timestep, images, n_points: int) -> List[torch.Tensor]:
    """n/4 k-medoid + n/3 Shi-Tomasi + the rest random points per mask, in that order (query_points.py:197-237).  The
    shipped default (configs/model/sam_pt.yaml: 1 negative point, method "mixed") degenerates to ONE RANDOM point; the
    Shi-Tomasi share only exists from n = 3 on and then needs cv2 (see ``extract_corner_points``)."""
    n_kmedoid, n_shi_tomasi = n_points // 4, n_points // 3
    n_random = n_points - n_kmedoid - n_shi_tomasi
    parts = []
    if n_kmedoid > 0:
        parts.append([extract_kmedoid_points(qm, n_kmedoid) for qm in query_masks])
    if n_shi_tomasi > 0:
        parts.append([extract_corner_points(images[int(t.item())], qm, n_shi_tomasi)
                      for qm, t in zip(query_masks, query_points_timestep)])
    if n_random > 0:
        parts.append([extract_random_mask_points(qm, n_random) for qm in query_masks])
    if len(parts) == 1:
        return parts[0]
    return [torch.cat(x, dim=0) for x in zip(*parts)]


def extract_query_points_xy(images, query_masks, query_points_timestep, method: str, points_per_mask: int) -> List[torch.Tensor]:
    """Dispatch of SamPt._extract_query_points_xy (sam_pt.py:290-306)."""
    if method == "kmedoids":
        return [extract_kmedoid_points(qm, points_per_mask) for qm in query_masks]
    if method == "random":
        return [extract_random_mask_points(qm, points_per_mask) for qm in query_masks]
    if method == "shi-tomasi":
        return [extract_corner_points(images[int(t.item())], qm, points_per_mask)
                for qm, t in zip(query_masks, query_points_timestep)]
    if method == "mixed":
        return extract_mixed_points(query_masks, query_points_timestep, images, points_per_mask)
    raise NotImplementedError(f"Point selection method {method} not implemented")
