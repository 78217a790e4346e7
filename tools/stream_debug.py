"""Diagnostic: run-to-run determinism of SamPt.forward on the reduced geometry over several clips of different lengths
(multi-chunk, ragged decode batches), with the decode workspace left alone / zeroed / poisoned (0xFF = NaN patterns) before
every decode call, with and without hipGraph replay.  python tools/stream_debug.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_pt_amd.point_tracker import PipsPointTracker      # noqa: E402
from sam_pt_amd.sam_predictor import SamHip, SamPredictor  # noqa: E402
from sam_pt_amd.sam_pt import SamPt                        # noqa: E402
from sam_pt_amd.synth import disc_queries, synthetic_clip  # noqa: E402
from sam_pt_amd.weights import SAM_CONFIGS, init_pips_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
cfg = SAM_CONFIGS["vit_test"]
psd = init_pips_state_dict(72)


def build(thr, graph, fill, sync="", fmax=8):
    pred = SamPredictor(SamHip(config=cfg, seed=72, precision="f32", max_batch=4, max_decode_batch=fmax).to(dev))
    pred.use_graph = graph
    if sync:
        orig_td = pred.track_decode

        def synced(*a, **k):
            if "before" in sync:
                torch.cuda.synchronize()
            r = orig_td(*a, **k)
            if "after" in sync:
                torch.cuda.synchronize()
            return r
        pred.track_decode = synced
    if fill is not None:
        orig = pred._dec_ws

        def filled(*a, **k):
            ws = orig(*a, **k)
            ws.fill_(fill)
            return ws
        pred._dec_ws = filled
    return SamPt(PipsPointTracker(state_dict=psd), pred, sam_iou_threshold=thr, positive_points_per_mask=4,
                 negative_points_per_mask=0, iterative_refinement_iterations=3).eval()


videos = []
for seed, T in ((72, 11), (73, 9), (74, 11), (75, 5)):
    frames, centres = synthetic_clip(T=T, H=128, W=256, seed=seed)
    q = torch.stack([disc_queries(centres, n_pos=4, r=9.0), disc_queries(centres, n_pos=4, r=5.0) + torch.tensor([0.0, -50.0, 20.0])])
    videos.append({"image": [f.to(dev) for f in frames], "target_hw": (128, 256), "query_points": q})


def diff(a, b):
    la, lb = torch.stack(a["logits"]), torch.stack(b["logits"])
    fin = torch.isfinite(la) & torch.isfinite(lb)
    sa, sb = torch.tensor(a["scores_per_frame"]), torch.tensor(b["scores_per_frame"])
    return (f"nan {int(torch.isnan(la).sum())}/{int(torch.isnan(lb).sum())} finite-pattern {bool(torch.equal(torch.isfinite(la), torch.isfinite(lb)))} "
            f"logits max|d| {float((la[fin] - lb[fin]).abs().max()) if fin.any() else -1:.3e} scores max|d| {float((sa - sb).abs().nan_to_num(0).max()):.3e}")


base = None
for graph, fill, sync, fmax in ((False, None, "", 8), (True, None, "", 8), (False, None, "", 32), (True, None, "", 32)):
    model = build(-1e9, graph, fill, sync, fmax)
    runs = []
    for rep in range(4):
        runs.append([model(v) for v in videos])
        torch.cuda.synchronize()
    if fmax == 32 and not graph:
        base = runs[0]
    print(f"== graph {graph} sync '{sync}' max_decode_batch {fmax}: runs 2 / 3 / 4 vs " + ("run 1" if base is None else "the eager baseline"),
          "graph stats", model.sam_predictor.graph_stats())
    for i in range(len(videos)):
        print("  clip", i, "  |  ".join(diff(r[i], (base or runs[0])[i]) for r in runs[1:]))
    if base is None:
        base = runs[0]
