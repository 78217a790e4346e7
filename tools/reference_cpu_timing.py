#!/usr/bin/env python
"""Build-container only: time the REFERENCE's own PipsPointTracker (sam_pt/point_tracker/pips/tracker.py, imported in
place from /root/reference) and the reference SamPt loop over the CPU oracle predictor on the benchmark clip
(24 x 576x1024, 8 points, 1 object, R = 12; seeded weights).  Results are recorded in BASELINE.md §3.
   PYTHONDONTWRITEBYTECODE=1 python tools/reference_cpu_timing.py [vit_b] [sam_frames]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import reference_loader as RL
from oracle import sam_ref as R
from sam_pt_amd.synth import bench_clip
from sam_pt_amd.weights import SAM_CONFIGS, init_pips_state_dict, init_sam_state_dict

variant = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
n_sam = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.set_num_threads(os.cpu_count())
Pips, RefTracker, _ = RL.load_pips()
RefSamPt = RL.load_sam_pt()
psd = init_pips_state_dict(72)
frames, qp = bench_clip(T=24, seed=72, n_pos=8)
with tempfile.TemporaryDirectory() as d:
    torch.save({"model_state_dict": psd}, os.path.join(d, "model-000000001.pth"))
    trk = RefTracker(checkpoint_path=d, stride=4, s=8, initial_next_frame_visibility_threshold=0.9).eval()
t0 = time.time()
with torch.no_grad():
    out = trk.evaluate_batch(frames[None], qp[0][None])
t_trk = time.time() - t0
print(f"reference PipsPointTracker, T=24, 576x1024, N=8, {os.cpu_count()} threads: {t_trk:.1f} s -> {24 / t_trk:.3f} fps (tracker alone)", flush=True)
cfg = SAM_CONFIGS[variant]
sd = init_sam_state_dict(cfg, 72)
pred = R.SamPredictorRef(sd, cfg)
model = RefSamPt(point_tracker=trk, sam_predictor=pred, sam_iou_threshold=-1e9, iterative_refinement_iterations=12,
                 positive_point_selection_method="kmedoids", negative_point_selection_method="mixed",
                 positive_points_per_mask=8, negative_points_per_mask=0, add_other_objects_positive_points_as_negative_points=True,
                 max_other_objects_positive_points=None, point_tracker_mask_batch_size=5, use_patch_matching_filtering=False,
                 patch_size=3, patch_similarity_threshold=0.01, use_point_reinit=False, reinit_point_tracker_horizon=24,
                 reinit_horizon=24, reinit_variant="reinit-at-median-of-area-diff").eval()
traj, vis = out["trajectories_pred"][0].reshape(24, 1, 8, 2), out["visibilities_pred"][0].reshape(24, 1, 8).float()
t0 = time.time()
with torch.no_grad():
    model._apply_sam_to_trajectories(frames[:n_sam], traj[:n_sam], vis[:n_sam])
t_sam = (time.time() - t0) / n_sam
print(f"reference SamPt._apply_sam_to_trajectories over the CPU oracle SAM ({variant}), {n_sam} frames: {t_sam:.1f} s per frame "
      f"(set_image + 13 predict_torch)", flush=True)
tot = t_trk + 24 * t_sam
print(f"=> reference code path end to end (tracker measured on the whole clip + 24 x SAM per-frame time): {tot:.0f} s per clip = {24 / tot:.4f} fps")
