"""Query-point selection from a mask (reference: sam_pt/utils/query_points.py).

Needed by ``SamPt`` in ``query_masks`` mode (every VOS run, sam_pt.py:171-177) and by point re-initialisation
(sam_pt.py:529-533), once per object per (re)initialisation on at most 1800 sub-sampled mask pixels.  The pixel draw (one
``torch.randperm`` on the global generator, as the reference consumes it) stays on the host; the k-medoid clustering — 60 to
160 ms of numpy per mask, inside the evaluator's timed window — runs on the HIP device when the caller passes one
(``kmedoids_alternate_device``, csrc/kmedoids.hip), bit-identical to the host restatement below.

* ``extract_random_mask_points`` — same RNG consumption as the reference (one ``torch.randperm`` on the global
  generator, query_points.py:55), so results are bit-identical for the same seed (pinned in tests).
* ``extract_kmedoid_points`` — the reference calls ``sklearn_extra.cluster.KMedoids(n_clusters=N).fit`` (defaults:
  euclidean metric, ``method='alternate'``, ``init='heuristic'``, ``max_iter=300``).  scikit-learn-extra is a
  third-party dependency that is absent here, so ``kmedoids_alternate`` restates its published algorithm; **parity
  unpinned** for this function.
* ``extract_mixed_points`` — the reference's n/4 k-medoid + n/3 Shi-Tomasi + rest random split; with the shipped default
  of one negative point it is a single random point.
* ``extract_corner_points`` — Shi-Tomasi corners on an eroded mask.  The reference uses ``cv2.cvtColor``, ``cv2.erode``
  and ``cv2.goodFeaturesToTrack``; OpenCV is absent, so its published algorithms are restated in numpy — **parity
  unpinned** against OpenCV itself (the erosion is pinned on ``scipy.ndimage.binary_erosion`` and the min-eigenvalue map
  cross-checked against ``scipy.ndimage`` filters, tests/test_cpu_host.py).  With a HIP device the whole corner selection runs
  there (``shi_tomasi_device``, csrc/corners.hip), bit-identical to the numpy restatement.
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


def kmedoids_alternate_device(X: torch.Tensor, n_clusters: int, device, max_iter: int = 300) -> np.ndarray:
    """``kmedoids_alternate`` on the HIP device, bit-identical to the host restatement (csrc/kmedoids.hip: fp64 distances, numpy's
    pairwise summation order, first-index ties).  X: (n, 2) float32 pixel coordinates on the host, n <= 2048.  The heuristic
    initialisation's ``np.argpartition`` stays on the host (its output order is numpy's introselect): the device returns the row
    sums, the host partitions n doubles, the device iterates to convergence — two small round trips instead of 60 - 160 ms of
    numpy on an n x n fp64 matrix."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = int(X.shape[0])
    with _lib.device_guard(torch.device(device)):
        xy = X.detach().to(torch.float32).contiguous().to(device)
        sums = torch.empty(n, dtype=torch.float64, device=device)
        _lib.check(lib.sampt_kmedoids_rowsums_f64(_lib.ptr(xy), n, _lib.ptr(sums), _lib.stream_ptr()), "sampt_kmedoids_rowsums_f64")
        medoids = np.argpartition(sums.cpu().numpy(), n_clusters - 1)[:n_clusters]           # heuristic init (host: introselect order)
        med = torch.from_numpy(np.ascontiguousarray(medoids.astype(np.int32))).to(device)
        _lib.check(lib.sampt_kmedoids_alternate(_lib.ptr(xy), n, int(n_clusters), _lib.ptr(med), int(max_iter), None,
                                                _lib.stream_ptr()), "sampt_kmedoids_alternate")
        return med.cpu().numpy().astype(np.int64)


def extract_kmedoid_points(mask: torch.Tensor, n_points_to_select: int, subsample_size: int = 1800, device=None) -> torch.Tensor:
    """K-medoid centres of (a random 1800-pixel subsample of) the mask, as (x, y) (query_points.py:62-99).  ``device``: a HIP
    device runs the clustering there (``kmedoids_alternate_device``: same result bit for bit); None = the numpy restatement."""
    if mask.sum() == 0:
        print("Warning: mask.sum() == 0 in extract_kmedoid_points")
        return torch.zeros((n_points_to_select, 2))
    px = mask.nonzero().float()
    if len(px) < n_points_to_select:
        sel = px.repeat(n_points_to_select // len(px) + 1, 1)[:n_points_to_select]
    else:
        px = px[torch.randperm(len(px))[:subsample_size]]
        if device is not None and torch.device(device).type == "cuda" and len(px) <= 2048 and n_points_to_select <= 64:
            idx = kmedoids_alternate_device(px, n_points_to_select, device)
        else:
            idx = kmedoids_alternate(px.numpy(), n_points_to_select)
        sel = px[torch.as_tensor(idx, dtype=torch.long)].type(torch.float32)
    return sel.flip(1)


# ---- Shi-Tomasi corners without OpenCV -------------------------------------------------------------------------------
# The reference calls cv2.cvtColor / cv2.erode / cv2.goodFeaturesToTrack (query_points.py:102-194).  OpenCV is a
# third-party dependency that is absent here, so the functions below restate its published algorithms (imgproc:
# color_rgb RGB2Gray, morph erode, corner.cpp cornerMinEigenVal, featureselect.cpp goodFeaturesToTrack).  PARITY UNPINNED:
# there is no OpenCV in this image to compare against; the property tests in tests/test_cpu_host.py only check the
# algorithm's invariants.
def _rgb_to_gray_u8(img: np.ndarray) -> np.ndarray:
    """(H,W,3) uint8 RGB -> uint8 gray with OpenCV's 15-bit fixed-point weights (0.299, 0.587, 0.114)."""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15).astype(np.uint8)


def _erode(mask: np.ndarray, k: int) -> np.ndarray:
    """cv2.erode(mask, np.ones((k, k))): anchor at (k//2, k//2), pixels outside the image never erode.  An empty kernel
    (k = 0) is OpenCV's default 3x3; k = 1 is the identity."""
    if k == 0:
        k = 3
    if k == 1:
        return mask.copy()
    H, W = mask.shape
    a = k // 2
    pad = np.ones((H + k - 1, W + k - 1), dtype=mask.dtype)
    pad[a:a + H, a:a + W] = mask
    out = np.ones_like(mask)
    for i in range(k):
        for j in range(k):
            out &= pad[i:i + H, j:j + W]
    return out


def erode_mask_proportional_to_its_furthest_points_distance(mask: torch.Tensor, erosion_percentage: float) -> torch.Tensor:
    """query_points.py:165-194: square erosion by a percentage of the mask's bounding-box diagonal."""
    px = mask.nonzero().float()
    diameter = torch.norm(px.max(0)[0] - px.min(0)[0]).item()
    k = int(diameter * erosion_percentage)
    return torch.from_numpy(_erode(mask.cpu().numpy().astype(np.uint8), k)).type(mask.dtype).to(mask.device)


def _reflect101(a: np.ndarray, p: int) -> np.ndarray:
    return np.pad(a, p, mode="reflect")


def corner_min_eigen_val(gray: np.ndarray, block_size: int = 3, ksize: int = 3) -> np.ndarray:
    """cv::cornerMinEigenVal on a uint8 image: Sobel derivatives scaled by 1/(2^(ksize-1) * block_size * 255), unnormalised
    box filter of (dx^2, dx dy, dy^2) over block_size, smaller eigenvalue; BORDER_REFLECT_101 throughout."""
    assert ksize == 3 and block_size == 3
    g = _reflect101(gray.astype(np.float32), 1)
    scale = np.float32(1.0 / (4.0 * block_size * 255.0))
    dx = ((g[:-2, 2:] - g[:-2, :-2]) + 2 * (g[1:-1, 2:] - g[1:-1, :-2]) + (g[2:, 2:] - g[2:, :-2])) * scale
    dy = ((g[2:, :-2] - g[:-2, :-2]) + 2 * (g[2:, 1:-1] - g[:-2, 1:-1]) + (g[2:, 2:] - g[:-2, 2:])) * scale

    def box(a):
        p = _reflect101(a.astype(np.float32), 1)
        H, W = a.shape
        return sum(p[i:i + H, j:j + W] for i in range(3) for j in range(3))

    a, b, c = box(dx * dx) * np.float32(0.5), box(dx * dy), box(dy * dy) * np.float32(0.5)
    return ((a + c) - np.sqrt((a - c) * (a - c) + b * b)).astype(np.float32)


def good_features_to_track(gray: np.ndarray, max_corners: int, quality_level: float, min_distance: float,
                           mask: np.ndarray) -> np.ndarray:
    """cv::goodFeaturesToTrack (Shi-Tomasi, blockSize 3, gradientSize 3): local maxima of the min-eigenvalue map above
    quality_level * max, strongest first, greedily thinned to a minimum mutual distance.  -> (n, 2) float32 (x, y)."""
    eig = corner_min_eigen_val(gray)
    H, W = eig.shape
    m = mask.astype(bool)
    if not m.any():
        return np.empty((0, 2), np.float32)
    max_val = eig[m].max()
    eig = np.where(eig > max_val * quality_level, eig, 0).astype(np.float32)         # THRESH_TOZERO
    p = np.pad(eig, 1, mode="constant", constant_values=-np.inf)
    dil = np.max([p[i:i + H, j:j + W] for i in range(3) for j in range(3)], axis=0)  # 3x3 dilation
    cand = (eig != 0) & (eig == dil) & m
    cand[0, :] = cand[-1, :] = False                                                # OpenCV scans 1 .. size-2 only
    cand[:, 0] = cand[:, -1] = False
    ys, xs = np.nonzero(cand)
    if len(ys) == 0:
        return np.empty((0, 2), np.float32)
    order = np.lexsort((-(ys * W + xs), -eig[ys, xs]))       # value descending, ties: higher address first
    ys, xs = ys[order], xs[order]
    if min_distance < 1:
        sel = list(range(min(len(ys), max_corners) if max_corners > 0 else len(ys)))
        return np.stack([xs[sel], ys[sel]], axis=1).astype(np.float32)
    cell = int(round(min_distance))
    gw, gh = (W + cell - 1) // cell, (H + cell - 1) // cell
    grid = [[] for _ in range(gw * gh)]
    md2 = min_distance * min_distance
    out = []
    for y, x in zip(ys.tolist(), xs.tolist()):
        cx, cy = x // cell, y // cell
        good = True
        for yy in range(max(0, cy - 1), min(gh - 1, cy + 1) + 1):
            for xx in range(max(0, cx - 1), min(gw - 1, cx + 1) + 1):
                for (px, py) in grid[yy * gw + xx]:
                    if (x - px) * (x - px) + (y - py) * (y - py) < md2:
                        good = False
                        break
                if not good:
                    break
            if not good:
                break
        if good:
            grid[cy * gw + cx].append((x, y))
            out.append((x, y))
            if 0 < max_corners <= len(out):
                break
    return np.asarray(out, dtype=np.float32).reshape(-1, 2)


def erode_device(mask: torch.Tensor, k: int, device) -> torch.Tensor:
    """``_erode`` on the HIP device (csrc/corners.hip, separable): mask (H,W) {0,1} -> uint8 (H,W) on ``device``."""
    from . import _lib
    lib = _lib.load()
    with _lib.device_guard(torch.device(device)):
        m = (mask > 0).to(torch.uint8).contiguous().to(device)
        tmp, out = torch.empty_like(m), torch.empty_like(m)
        _lib.check(lib.sampt_qp_erode_u8(_lib.ptr(m), m.shape[0], m.shape[1], int(k), _lib.ptr(tmp), _lib.ptr(out), _lib.stream_ptr()),
                   "sampt_qp_erode_u8")
        return out


def shi_tomasi_device(image: torch.Tensor, mask: torch.Tensor, n_points: int, device, quality_level: float = 0.001):
    """The Shi-Tomasi half of ``extract_corner_points`` on the HIP device (csrc/corners.hip): gray conversion, the 6 % / 2 % / 1 %
    erosion cascade, min-eigenvalue map, threshold, local maxima and the greedy minimum-distance selection in ONE launch sequence
    without a host round trip; one download of (n, 2) corners + 16 ints at the end.  Bit-identical to the host functions above
    (tests/test_gpu_kernels.py::test_shi_tomasi_device_equals_host_restatement).  -> (corners (n_found, 2) float32 (x, y) on the
    host, info dict: k of the erosion kept (-1 = the mask itself), eroded pixel count / bounding box)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    H, W = int(mask.shape[0]), int(mask.shape[1])
    with _lib.device_guard(torch.device(device)):
        img = image.to(device=device, dtype=torch.uint8).contiguous()
        m = (mask > 0).to(torch.uint8).contiguous().to(device)
        nb = C.c_size_t()
        _lib.check(lib.sampt_qp_corners_workspace_bytes(H, W, C.byref(nb)), "sampt_qp_corners_workspace_bytes")
        ws = torch.empty(nb.value, dtype=torch.uint8, device=device)
        xy = torch.zeros((n_points, 2), dtype=torch.float32, device=device)
        info = torch.zeros(16, dtype=torch.int32, device=device)
        _lib.check(lib.sampt_qp_shi_tomasi(_lib.ptr(img), _lib.ptr(m), H, W, int(n_points), float(quality_level), _lib.ptr(xy),
                                           _lib.ptr(info), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "sampt_qp_shi_tomasi")
        inf = info.cpu().tolist()                                  # the one synchronisation
        return xy.cpu()[:inf[12]], {"k": inf[10], "eroded_pixels": inf[9], "eroded_bbox": inf[5:9], "mask_bbox": inf[0:4],
                                    "mask_pixels": inf[4], "candidates": inf[14]}


def extract_corner_points(image: torch.Tensor, mask: torch.Tensor, n_points_to_select: int,
                          kmedoid_subsample_size: int = 2000, device=None) -> torch.Tensor:
    """Shi-Tomasi corners inside the eroded mask, topped up with k-medoid points (query_points.py:102-162).
    image (3,H,W) uint8, mask (H,W) {0,1} -> (n,2) float32 (x, y).  ``device``: a HIP device runs the corner selection there
    (``shi_tomasi_device``: the same corners bit for bit).  PARITY UNPINNED against OpenCV (see the note above)."""
    if mask.sum() == 0:
        print("Warning: mask.sum() == 0 in extract_corner_points")
        return torch.zeros((n_points_to_select, 2))
    if (device is not None and torch.device(device).type == "cuda" and 1 <= n_points_to_select <= 64
            and min(mask.shape[-2:]) >= 3):
        corners, _ = shi_tomasi_device(image, mask, n_points_to_select, device)
        if len(corners) < n_points_to_select:
            corners = torch.cat((corners, extract_kmedoid_points(mask.cpu(), n_points_to_select - corners.shape[0],
                                                                 subsample_size=kmedoid_subsample_size, device=device)), dim=0)
        assert corners.shape == (n_points_to_select, 2)
        return corners
    img = image.permute(1, 2, 0).cpu().numpy()
    eroded = erode_mask_proportional_to_its_furthest_points_distance(mask, 0.06)
    for pct in (0.02, 0.01):
        if eroded.sum() < 10:
            eroded = erode_mask_proportional_to_its_furthest_points_distance(mask, pct)
    if eroded.sum() < 10:
        eroded = mask
    px = eroded.nonzero().float()
    diameter = torch.norm(px.max(0)[0] - px.min(0)[0]).item()
    corners = good_features_to_track(_rgb_to_gray_u8(img), n_points_to_select, 0.001, diameter / n_points_to_select,
                                     eroded.cpu().numpy().astype(np.uint8))
    corners = torch.from_numpy(corners).type(torch.float32)
    if len(corners) < n_points_to_select:
        corners = torch.cat((corners, extract_kmedoid_points(mask, n_points_to_select - corners.shape[0],
                                                             subsample_size=kmedoid_subsample_size, device=device)), dim=0)
    assert corners.shape == (n_points_to_select, 2)
    return corners


def extract_mixed_points(query_masks, query_points_timestep, images, n_points: int, device=None) -> List[torch.Tensor]:
    """n/4 k-medoid + n/3 Shi-Tomasi + the rest random points per mask, in that order (query_points.py:197-237).  The
    shipped default (configs/model/sam_pt.yaml: 1 negative point, method "mixed") degenerates to ONE RANDOM point; the
    Shi-Tomasi share only exists from n = 3 on (see ``extract_corner_points`` for its parity status)."""
    n_kmedoid, n_shi_tomasi = n_points // 4, n_points // 3
    n_random = n_points - n_kmedoid - n_shi_tomasi
    parts = []
    if n_kmedoid > 0:
        parts.append([extract_kmedoid_points(qm, n_kmedoid, device=device) for qm in query_masks])
    if n_shi_tomasi > 0:
        parts.append([extract_corner_points(images[int(t.item())], qm, n_shi_tomasi, device=device)
                      for qm, t in zip(query_masks, query_points_timestep)])
    if n_random > 0:
        parts.append([extract_random_mask_points(qm, n_random) for qm in query_masks])
    if len(parts) == 1:
        return parts[0]
    return [torch.cat(x, dim=0) for x in zip(*parts)]


def extract_query_points_xy(images, query_masks, query_points_timestep, method: str, points_per_mask: int,
                            device=None) -> List[torch.Tensor]:
    """Dispatch of SamPt._extract_query_points_xy (sam_pt.py:290-306).  ``device``: run the k-medoid clustering on that HIP
    device (same points, bit for bit)."""
    if method == "kmedoids":
        return [extract_kmedoid_points(qm, points_per_mask, device=device) for qm in query_masks]
    if method == "random":
        return [extract_random_mask_points(qm, points_per_mask) for qm in query_masks]
    if method == "shi-tomasi":
        return [extract_corner_points(images[int(t.item())], qm, points_per_mask, device=device)
                for qm, t in zip(query_masks, query_points_timestep)]
    if method == "mixed":
        return extract_mixed_points(query_masks, query_points_timestep, images, points_per_mask, device=device)
    raise NotImplementedError(f"Point selection method {method} not implemented")
