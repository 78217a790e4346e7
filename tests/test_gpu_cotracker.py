"""GPU parity tests for the CoTracker path: ``sam_pt_amd.CoTrackerPointTracker`` (HIP engine behind
the reference adapter's constructor, cotracker/tracker.py:27-170) against ``oracle/cotracker_ref.py`` on the same seeded
weights and inputs.  The oracle's MODEL half is **parity unpinned** (third-party co-tracker @ 4f297a9 is absent and has no
independent implementation here, see the oracle's header); its adapter half restates in-tree reference code.
Bars: visibilities identical, trajectories identical in index space (``round``) and within 5e-3 px."""
import numpy as np
import pytest
import torch

from tests.util import disc_queries, iou, max_abs, synthetic_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cot_sd():
    from sam_pt_amd.weights import init_cotracker_state_dict
    return init_cotracker_state_dict(72)


def _check(trk, ref, rgbs, q, dev, tol=5e-3):
    tr_ref, vi_ref = ref.forward(rgbs, q)
    out = trk.evaluate_batch(rgbs.to(dev), q.to(dev))
    tr, vi = out["trajectories_pred"], out["visibilities_pred"]
    assert tr.shape == tr_ref.shape and vi.shape == vi_ref.shape
    err = max_abs(tr, tr_ref)
    print(f"\n[cotracker] T={rgbs.shape[1]} N={q.shape[1]} max|dtraj|={err:.2e} px, visible {vi_ref.float().mean():.2f}")
    assert (vi.bool() == vi_ref).all(), "visibilities differ"
    assert err < tol
    assert (tr.round() == tr_ref.round()).all(), "trajectories differ in index space"
    return tr, vi


@pytest.mark.parametrize("T", [14, 9, 5])
def test_cotracker_tracker_vs_oracle(dev, cot_sd, T):
    """Queries on several frames (incl. the last one), support grid every 12 frames, both temporal directions and the
    `== 0` back-fill; T = 5 exercises the short-video padding of tracker.py:17-21, T = 9 a clip with a padded tail."""
    from oracle.cotracker_ref import CoTrackerTrackerRef
    from sam_pt_amd.point_tracker import CoTrackerPointTracker
    frames, centres = synthetic_clip(T=T, H=128, W=256, seed=72)
    q = torch.cat([disc_queries(centres, n_pos=4, r=9.0, t=0), disc_queries(centres, n_pos=2, r=6.0, t=min(5, T - 2)),
                   disc_queries(centres, n_pos=1, r=3.0, t=T - 1)])[None]
    kw = dict(interp_shape=(96, 128), visibility_threshold=0.7, support_grid_size=2, support_grid_every_n_frames=12)
    _check(CoTrackerPointTracker(state_dict=cot_sd, **kw), CoTrackerTrackerRef(cot_sd, **kw), frames[None], q, dev)


def test_cotracker_default_geometry_vs_oracle(dev, cot_sd):
    """The shipped configuration (configs/model/point_tracker/cotracker.yaml: 384x512, grid 2 every 12 frames) on frames of
    the bench geometry (576x1024), 8 positive + 8 negative queries (BASELINE config #3's prompt shape), T = 13."""
    from oracle.cotracker_ref import CoTrackerTrackerRef
    from sam_pt_amd.point_tracker import CoTrackerPointTracker
    from sam_pt_amd.synth import bench_clip
    frames, qp = bench_clip(T=13, seed=72, n_pos=16)
    q = qp[0][None]
    trk = CoTrackerPointTracker(state_dict=cot_sd)
    tr, vi = _check(trk, CoTrackerTrackerRef(cot_sd), frames[None], q, dev, tol=1e-2)
    assert trk.stats["fnet_frames"] == 13 and trk.stats["calls"] == 2          # every frame encoded once, 2 directions


def test_cotracker_long_clip_default_head_within_the_oracles_own_noise(dev, cot_sd):
    """BASELINE config #3's tracker shape — 8 + 8 points, T = 50: 12 chained windows per direction — with the DEFAULT flow
    head (x 0.003), the workload bench.py's config-#3 line times.  Over that many windows the random-weight model amplifies
    round-off until index identity is not a property any implementation can have (the conditioned workloads of
    oracle/workloads.py avoid the regime; the strict bar on this head is held at T = 13 by
    ``test_cotracker_default_geometry_vs_oracle``).  This test makes that statement evidence instead of an excuse: the
    oracle's distance to ITSELF under a 1e-7 relative weight perturbation (oracle/noise_floor.py) is measured beside the
    HIP-vs-oracle distance, and the device result must be as close to the oracle as the oracle is to itself (within 3 x the
    floor — or 0.75 px, should the single perturbation drawn here land unluckily close — visibilities and rounded indices
    differing on about as many entries).  Both distances are printed — the record VERDICT r4 asked for."""
    from oracle.cotracker_ref import CoTrackerTrackerRef
    from oracle.noise_floor import distance, tracker_noise_floor
    from sam_pt_amd.point_tracker import CoTrackerPointTracker
    from sam_pt_amd.synth import bench_clip
    torch.set_num_threads(min(32, torch.get_num_threads()))
    frames, qp = bench_clip(T=50, seed=72, n_pos=8, n_neg=8)
    q = qp[0][None]
    nf = tracker_noise_floor(lambda sd: CoTrackerTrackerRef(sd), cot_sd, frames[None], q)
    out = CoTrackerPointTracker(state_dict=cot_sd).evaluate_batch(frames[None].to(dev), q.to(dev))
    hip = distance(out["trajectories_pred"], out["visibilities_pred"], *nf["a"])
    floor = nf["floor"]
    print(f"\n[cotracker T=50 default head] oracle vs perturbed oracle (rel 1e-7): {floor}\n"
          f"[cotracker T=50 default head] HIP vs oracle:                          {hip}")
    # (one perturbation is ONE draw from a chaotic map, and the oracle's own rounding depends on the host's thread count: the
    #  bounds leave room for an unluckily small draw — measured on three boxes: floor 0.276 px / 22 / 2, HIP 0.222 px / 21 / 3)
    assert hip["traj_max_abs_px"] <= max(3.0 * floor["traj_max_abs_px"], 0.75), (hip, floor)
    assert hip["vis_differing"] <= 3 * floor["vis_differing"] + 8, (hip, floor)
    assert hip["traj_index_differing"] <= 3 * floor["traj_index_differing"] + 40, (hip, floor)
    assert floor["traj_max_abs_px"] > 1e-3, "the default head over 12 chained windows is expected to amplify a 1e-7 perturbation"


def test_cotracker_no_grid_native_shape(dev, cot_sd):
    """support_grid_size = 0 and interp_shape = None (the model runs at the frame size; tracker.py:87-88)."""
    from oracle.cotracker_ref import CoTrackerTrackerRef
    from sam_pt_amd.point_tracker import CoTrackerPointTracker
    frames, centres = synthetic_clip(T=10, H=128, W=256, seed=3)
    q = disc_queries(centres, n_pos=3, r=8.0, t=4)[None]
    kw = dict(interp_shape=None, visibility_threshold=0.7, support_grid_size=0, support_grid_every_n_frames=12)
    trk = CoTrackerPointTracker(state_dict=cot_sd, **kw)
    _check(trk, CoTrackerTrackerRef(cot_sd, **kw), frames[None], q, dev)
    assert trk.interp_shape == (128, 256)


def test_sampt_with_cotracker_end_to_end_vs_oracle(dev, cot_sd):
    """SamPt.forward with the CoTracker seam (the reference's default tracker, configs/model/sam_pt.yaml:4): fused device
    path vs the reference protocol driven on the CPU oracle (positives + negatives: two prompt passes per frame)."""
    from oracle.cotracker_ref import CoTrackerTrackerRef
    from oracle.parity import compare, reference_run
    from sam_pt_amd.point_tracker import CoTrackerPointTracker
    from sam_pt_amd.sam_predictor import SamHip, SamPredictor
    from sam_pt_amd.sam_pt import SamPt
    from sam_pt_amd.weights import SAM_CONFIGS, init_sam_state_dict
    cfg = SAM_CONFIGS["vit_test"]
    sd = init_sam_state_dict(cfg, 72)
    frames, centres = synthetic_clip(T=10, H=128, W=256, seed=72)
    q = disc_queries(centres, n_pos=6, r=9.0)
    q[4:, 1:] += torch.tensor([40.0, 30.0])
    kw = dict(sam_iou_threshold=-1e9, positive_points_per_mask=4, negative_points_per_mask=2,
              iterative_refinement_iterations=3, point_tracker_mask_batch_size=5)
    tkw = dict(interp_shape=(96, 128), visibility_threshold=0.7, support_grid_size=2, support_grid_every_n_frames=12)
    ref = reference_run(cfg, sd, None, frames, q[None], kw, tracker_factory=lambda: CoTrackerTrackerRef(cot_sd, **tkw))
    model = SamPt(CoTrackerPointTracker(state_dict=cot_sd, **tkw),
                  SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").to(dev)), **kw).eval()
    out = model({"image": [f.to(dev) for f in frames], "target_hw": (128, 256), "query_points": q[None]})
    res = compare(out, ref)
    print(f"\n[sampt+cotracker] {res}")
    assert res["vis_identical"] and res["traj_index_identical"] and res["rejections_identical"], res
    assert res["mask_iou_min"] >= 1 - 1e-3, res
