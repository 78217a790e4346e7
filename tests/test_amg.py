"""CPU tests of the automatic mask generator and the VIS adapter (no GPU needed).

The generator's helpers are pinned against transformers' independent port of segment-anything's ``utils/amg.py``; the
generator itself runs on the CPU oracle predictor; NMS and the small-region clean-up (torchvision / OpenCV upstream, both
absent here) are checked against brute-force definitions."""
import numpy as np
import pytest
import torch

from sam_pt_amd import automatic_mask_generator as A


def _hf():
    # the PIL-backend module: same helpers, and it imports without torchvision
    return pytest.importorskip("transformers.models.sam.image_processing_pil_sam")


def test_geometry_helpers_vs_transformers():
    H = _hf()
    for n in (1, 2, 5, 32):
        assert np.allclose(A.build_point_grid(n), np.asarray(H._build_point_grid(n)), atol=1e-6)
    grids = A.build_all_layer_point_grids(32, 2, 2)
    assert [len(g) for g in grids] == [1024, 256, 64]
    for size in ((480, 854), (576, 1024), (333, 500), (1024, 1024)):
        for layers in (0, 1, 2):
            ours = A.generate_crop_boxes(size, layers, 512 / 1500)
            theirs = H._generate_per_layer_crops(layers, 512 / 1500, size)
            assert ours[0] == theirs[0] and ours[1] == theirs[1]
            assert len(ours[0]) == sum(4 ** i for i in range(layers + 1))


def test_mask_helpers_vs_transformers():
    H = _hf()
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(7, 40, 56, generator=g) * 3
    logits[3] = -5.0                                                    # empty after thresholding
    logits[4, 10:20, 5:50] = 9.0
    for thr, off in ((0.0, 1.0), (0.5, 0.25)):
        ours = A.calculate_stability_score(logits, thr, off)
        theirs = H._compute_stability_score(logits, thr, off)
        assert torch.equal(torch.nan_to_num(ours, nan=-1.0), torch.nan_to_num(theirs, nan=-1.0))
    masks = logits > 0
    assert torch.equal(A.batched_mask_to_box(masks), H._batched_mask_to_box(masks))
    assert A.batched_mask_to_box(masks)[3].tolist() == [0, 0, 0, 0]
    m4 = masks.reshape(1, 7, 40, 56)
    assert torch.equal(A.batched_mask_to_box(m4), H._batched_mask_to_box(m4))
    boxes = torch.tensor([[0, 0, 30, 30], [5, 40, 55, 80], [100, 3, 199, 100], [30, 30, 60, 60], [0, 25, 199, 99]])
    for crop, orig in (([0, 0, 200, 100], [0, 0, 200, 100]), ([50, 20, 250, 120], [0, 0, 400, 300]),
                       ([0, 20, 200, 120], [0, 0, 400, 120])):
        assert torch.equal(A.is_box_near_crop_edge(boxes, crop, orig), H._is_box_near_crop_edge(boxes, crop, orig))
    for crop in ([0, 0, 56, 40], [10, 5, 66, 45], [0, 7, 56, 47]):
        assert torch.equal(A.uncrop_masks(masks, crop, 60, 80) if crop != [0, 0, 56, 40] else A.uncrop_masks(masks, crop, 40, 56),
                           H._pad_masks(masks, crop, 60, 80) if crop != [0, 0, 56, 40] else H._pad_masks(masks, crop, 40, 56))
    full = torch.ones(1, 40, 56, dtype=torch.bool)
    allm = torch.cat([masks, full, ~full])
    ours, theirs = A.mask_to_rle(allm), H._mask_to_rle(allm)
    for a, b, m in zip(ours, theirs, allm):
        assert a["size"] == b["size"] and [int(c) for c in a["counts"]] == [int(c) for c in b["counts"]]
        assert np.array_equal(A.rle_to_mask(a), m.numpy()) and np.array_equal(np.asarray(H._rle_to_mask(a)), m.numpy())
        assert A.area_from_rle(a) == int(m.sum())
    assert A.box_xyxy_to_xywh([3, 4, 10, 20]) == [3, 4, 7, 16]
    assert A.uncrop_boxes_xyxy(boxes[:1], [7, 9, 0, 0]).tolist() == [[7, 9, 37, 39]]
    assert A.uncrop_points(torch.tensor([[1.5, 2.0]]), [7, 9, 0, 0]).tolist() == [[8.5, 11.0]]


def _nms_bruteforce(boxes, scores, thr):
    order = sorted(range(len(boxes)), key=lambda i: (-scores[i], i))
    keep = []
    for i in order:
        ok = True
        for j in keep:
            x0, y0 = max(boxes[i][0], boxes[j][0]), max(boxes[i][1], boxes[j][1])
            x1, y1 = min(boxes[i][2], boxes[j][2]), min(boxes[i][3], boxes[j][3])
            inter = max(x1 - x0, 0) * max(y1 - y0, 0)
            ai = (boxes[i][2] - boxes[i][0]) * (boxes[i][3] - boxes[i][1])
            aj = (boxes[j][2] - boxes[j][0]) * (boxes[j][3] - boxes[j][1])
            union = ai + aj - inter
            if union > 0 and inter / union > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


def test_nms_vs_bruteforce():
    rng = np.random.default_rng(4)
    for n in (0, 1, 40, 300):
        xy = rng.integers(0, 80, size=(n, 2))
        wh = rng.integers(0, 40, size=(n, 2))                            # includes degenerate (zero-area) boxes
        boxes = np.concatenate([xy, xy + wh], axis=1).astype(np.float32)
        scores = rng.random(n).astype(np.float32)
        if n >= 40:
            boxes[5], scores[5] = boxes[3], scores[3]                    # exact duplicate with a tied score
        for thr in (0.3, 0.7):
            got = A.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).tolist()
            assert got == _nms_bruteforce(boxes.tolist(), scores.tolist(), thr)
            if n:
                iou = A.box_iou_matrix(torch.from_numpy(boxes[got]))
                iou.fill_diagonal_(0)
                assert not bool((torch.nan_to_num(iou) > thr).any())


def test_remove_small_regions():
    m = np.zeros((12, 16), dtype=bool)
    m[1:9, 1:9] = True                     # 64-px square
    m[4, 4] = False                        # 1-px hole
    m[10, 14] = True                       # 1-px island, diagonal neighbours only with nothing
    m[9, 9] = True                         # touches the square diagonally -> same 8-connected component
    out, changed = A.remove_small_regions(m, 4, "holes")
    assert changed and out[4, 4] and out.sum() == m.sum() + 1
    out2, changed2 = A.remove_small_regions(out, 4, "islands")
    assert changed2 and not out2[10, 14] and out2[9, 9] and out2[1:9, 1:9].all()
    same, changed3 = A.remove_small_regions(out2, 4, "islands")
    assert not changed3 and np.array_equal(same, out2)
    tiny = np.zeros((6, 6), dtype=bool)
    tiny[0, 0] = tiny[5, 4] = tiny[5, 5] = True
    kept, ch = A.remove_small_regions(tiny, 10, "islands")               # everything is small: the largest survives
    assert ch and kept.sum() == 2 and kept[5, 4] and kept[5, 5]


@pytest.fixture(scope="module")
def oracle_predictor():
    from oracle import sam_ref as R
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    return R.SamPredictorRef(init_sam_state_dict(cfg, 72), cfg)


def _image(h=96, w=128, seed=3):
    from sam_pt_amd.synth import synthetic_clip
    frames, _ = synthetic_clip(T=1, H=h, W=w, seed=seed)
    return frames[0].permute(1, 2, 0).contiguous().numpy()


def test_generator_records_on_oracle_predictor(oracle_predictor):
    img = _image()
    gen = A.SamAutomaticMaskGenerator(None, points_per_side=4, points_per_batch=5, pred_iou_thresh=0.0,
                                      stability_score_thresh=0.0, box_nms_thresh=0.7, predictor=oracle_predictor)
    recs = gen.generate(img)
    assert 0 < len(recs) <= 16 * 3
    ious = [r["predicted_iou"] for r in recs]
    assert ious == sorted(ious, reverse=True)                              # NMS order = decreasing predicted IoU
    oracle_predictor.set_image(img)
    grid = A.build_point_grid(4) * np.array([[128, 96]])
    for r in recs[:6]:
        m = r["segmentation"]
        assert m.shape == (96, 128) and m.dtype == bool and r["area"] == int(m.sum())
        x, y, w, h = r["bbox"]
        if m.any():
            ys, xs = np.nonzero(m)
            assert [x, y, x + w, y + h] == [xs.min(), ys.min(), xs.max(), ys.max()]
        assert r["crop_box"] == [0, 0, 128, 96]
        (px, py), = r["point_coords"]
        assert np.abs(grid - np.array([[px, py]])).sum(axis=1).min() < 1e-9
        # the record is one of the three masks SAM returns for exactly that point
        pc = torch.as_tensor(oracle_predictor.transform.apply_coords(np.array([[px, py]]), (96, 128)))[None].float()
        logits, iou, _ = oracle_predictor.predict_torch(pc, torch.ones(1, 1, dtype=torch.int), multimask_output=True,
                                                        return_logits=True)
        j = int(np.argmin(np.abs(iou[0].numpy() - r["predicted_iou"])))
        assert np.array_equal((logits[0, j] > 0).numpy(), m)
        st = A.calculate_stability_score(logits[0, j][None], 0.0, 1.0)[0]
        assert abs(float(st) - r["stability_score"]) < 1e-6 or (np.isnan(float(st)) and np.isnan(r["stability_score"]))
    boxes = torch.tensor([[r["bbox"][0], r["bbox"][1], r["bbox"][0] + r["bbox"][2], r["bbox"][1] + r["bbox"][3]]
                          for r in recs], dtype=torch.float)
    iou_m = torch.nan_to_num(A.box_iou_matrix(boxes))
    iou_m.fill_diagonal_(0)
    assert not bool((iou_m > 0.7).any())
    # thresholds only remove records; RLE output is the same masks
    strict = A.SamAutomaticMaskGenerator(None, points_per_side=4, points_per_batch=16, pred_iou_thresh=float(np.median(ious)),
                                         stability_score_thresh=0.0, predictor=oracle_predictor).generate(img)
    assert all(r["predicted_iou"] > float(np.median(ious)) for r in strict) and len(strict) < len(recs)
    rle = A.SamAutomaticMaskGenerator(None, points_per_side=4, points_per_batch=64, pred_iou_thresh=0.0,
                                      stability_score_thresh=0.0, output_mode="uncompressed_rle",
                                      predictor=oracle_predictor).generate(img)
    assert len(rle) == len(recs)
    assert all(np.array_equal(A.rle_to_mask(a["segmentation"]), b["segmentation"]) for a, b in zip(rle, recs))


def test_generator_crops_and_small_regions(oracle_predictor):
    img = _image(96, 128, seed=5)
    gen = A.SamAutomaticMaskGenerator(None, points_per_side=2, points_per_batch=8, pred_iou_thresh=0.0,
                                      stability_score_thresh=0.0, crop_n_layers=1, crop_n_points_downscale_factor=2,
                                      min_mask_region_area=6, predictor=oracle_predictor)
    recs = gen.generate(img)
    assert recs and all(r["segmentation"].shape == (96, 128) for r in recs)
    crops = {tuple(r["crop_box"]) for r in recs}
    assert crops <= {tuple(A.box_xyxy_to_xywh(b)) for b in A.generate_crop_boxes((96, 128), 1, 512 / 1500)[0]}
    for r in recs:                                                        # clean-up: no foreground island below 6 px
        from scipy import ndimage
        lab, n = ndimage.label(r["segmentation"], structure=np.ones((3, 3), dtype=bool))
        sizes = np.bincount(lab.ravel())[1:]
        assert n <= 1 or sizes.min() >= 6
        (px, py), = r["point_coords"]
        cx, cy, cw, ch = r["crop_box"]
        assert cx <= px <= cx + cw and cy <= py <= cy + ch
    with pytest.raises(ValueError):
        A.SamAutomaticMaskGenerator(None, points_per_side=None, point_grids=None, predictor=oracle_predictor)
    with pytest.raises(NotImplementedError):
        A.SamAutomaticMaskGenerator(None, output_mode="coco_rle", predictor=oracle_predictor)


def test_vis_adapter_contract(oracle_predictor):
    """SamBasedVisToVosAdapter.forward: proposals of frame 0 -> query masks in batches -> result dict
    (vis_to_vos_adapter.py:64-159), with a stub VOS model that records what it is asked."""
    from sam_pt_amd.vis_to_vos_adapter import SamBasedVisToVosAdapter
    T, Hh, Ww = 3, 96, 128
    frames = [torch.as_tensor(_image(Hh, Ww, seed=7 + t)).permute(2, 0, 1).contiguous() for t in range(T)]
    gen = A.SamAutomaticMaskGenerator(None, points_per_side=3, points_per_batch=9, pred_iou_thresh=0.0,
                                      stability_score_thresh=0.0, predictor=oracle_predictor)
    expect = gen.generate(frames[0].permute(1, 2, 0).numpy())
    calls = []

    class StubVos(torch.nn.Module):
        def forward(self, video):
            qm = video["query_masks"]
            calls.append((qm.shape[0], video["query_point_timestep"].tolist(), tuple(video["target_hw"])))
            M = qm.shape[0]
            logits = [qm[m].float()[None].repeat(T, 1, 1) * 2 - 1 for m in range(M)]
            return {"logits": logits, "trajectories": torch.zeros(T, M, 4, 2), "visibilities": torch.ones(T, M, 4),
                    "scores": [0.5 + 0.01 * m for m in range(M)]}

    n_keep = min(5, len(expect))
    adapter = SamBasedVisToVosAdapter(StubVos(), gen, max_num_masks=n_keep, masks_batch_size=2, visualize_results=True,
                                      max_videos_to_visualize=3)
    out = adapter([{"video_id": 0, "image": frames, "height": Hh, "width": Ww}])
    assert [c[0] for c in calls] == [2] * (n_keep // 2) + ([1] if n_keep % 2 else [])
    assert all(c[1] == [0] * c[0] and c[2] == (Hh, Ww) for c in calls)
    assert out["image_size"] == (Hh, Ww) and out["pred_labels"] == [0] * n_keep and len(out["pred_scores"]) == n_keep
    assert len(out["pred_masks"]) == n_keep and out["pred_masks"][0].shape == (T, Hh, Ww)
    for i in range(n_keep):
        assert np.array_equal(out["pred_masks"][i][0].numpy(), expect[i]["segmentation"])
    assert out["trajectories"].shape == (T, n_keep, 4, 2) and out["visibilities"].shape == (T, n_keep, 4)
    with pytest.raises(AssertionError):
        adapter([{"video_id": 0, "image": [f.float() for f in frames], "height": Hh, "width": Ww}])
