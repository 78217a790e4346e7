"""CPU tests of the CoTracker row (a13): oracle structure (the model half is parity unpinned, see oracle/cotracker_ref.py),
the adapter logic restated from sam_pt/point_tracker/cotracker/tracker.py, host-side packing, and the drop-in constructor."""
import os

import numpy as np
import pytest
import torch

from tests.util import disc_queries, synthetic_clip

REF_CFG = "/root/reference/configs/model/point_tracker/cotracker.yaml"


@pytest.fixture(scope="module")
def cot_sd():
    from sam_pt_amd.weights import init_cotracker_state_dict
    return init_cotracker_state_dict(72)


def test_state_dict_layout(cot_sd):
    """Checkpoint key layout of cotracker_stride_4_wind_8.pth: CoTracker(stride 4, S 8, 6 + 6 blocks)."""
    keys = set(cot_sd)
    for kind in ("time_blocks", "space_blocks"):
        for i in range(6):
            for leaf, shape in (("attn.qkv.weight", (1152, 384)), ("attn.proj.weight", (384, 384)),
                                ("mlp.fc1.weight", (1536, 384)), ("mlp.fc2.weight", (384, 1536))):
                assert tuple(cot_sd[f"updateformer.{kind}.{i}.{leaf}"].shape) == shape
    assert tuple(cot_sd["updateformer.input_transform.weight"].shape) == (384, 456)
    assert tuple(cot_sd["updateformer.flow_head.weight"].shape) == (130, 384)
    assert {"norm.weight", "norm.bias", "ffeat_updater.0.weight", "vis_predictor.0.weight", "fnet.conv1.weight",
            "fnet.layer2.0.downsample.0.weight", "fnet.conv2.weight", "fnet.conv3.bias"} <= keys
    assert not any("norm1" in k or "norm2" in k for k in keys if k.startswith("updateformer"))   # affine-free LayerNorms
    assert sum(v.numel() for v in cot_sd.values()) == 24149859


def test_support_grid_and_embeddings():
    from oracle import cotracker_ref as CR
    from sam_pt_amd.pack import cotracker_pos_tables, sincos_1d
    from sam_pt_amd.point_tracker import get_points_on_a_grid
    g = get_points_on_a_grid(2, (384, 512))
    assert torch.equal(g, torch.tensor([[8.0, 8.0], [504.0, 8.0], [8.0, 376.0], [504.0, 376.0]]))
    assert torch.equal(g, CR.get_points_on_a_grid(2, (384, 512)))
    assert torch.equal(get_points_on_a_grid(1, (384, 512)), torch.tensor([[256.0, 192.0]]))
    grid = CR.sincos_2d_grid(456, 24, 32)                       # product tables == the oracle's 2-D grid, factorised
    px, py = cotracker_pos_tables(24, 32)
    assert torch.equal(grid[5, 7, :228], px[7]) and torch.equal(grid[5, 7, 228:], py[5])
    assert torch.equal(sincos_1d(456, np.linspace(0, 7, 8)), torch.from_numpy(CR.sincos_1d(456, np.linspace(0, 7, 8))).float())
    e = CR.flow_embedding(torch.tensor([[0.5, -1.25]]))
    assert e.shape == (1, 130) and torch.equal(e[0, :2], torch.tensor([0.5, -1.25]))
    assert float(e[0, 2]) == 0.0 and float(e[0, 3]) == 1.0      # sin(0), cos(0): frequency 0 first, sin/cos interleaved


def test_oracle_adapter_semantics(cot_sd):
    """Adapter behaviour restated from tracker.py: result shapes, support points dropped, the `== 0` back-fill covers only
    the frames before the first window holding the query frame, clips shorter than the window work, visibilities are
    thresholded at 0.7, and the per-frame encoder cache serves both temporal directions."""
    from oracle import cotracker_ref as CR
    frames, centres = synthetic_clip(T=14, H=128, W=256, seed=72)
    q = torch.cat([disc_queries(centres, n_pos=2, r=9.0, t=0), disc_queries(centres, n_pos=1, r=3.0, t=13)])[None]
    trk = CR.CoTrackerTrackerRef(cot_sd, interp_shape=(96, 128))
    traj, vis = trk.forward(frames[None], q)
    assert traj.shape == (1, 14, 3, 2) and vis.shape == (1, 14, 3) and vis.dtype == torch.bool
    assert (traj != 0).all()                                    # every frame of every point filled by one of the passes
    assert trk.n_windows == 6                                   # 3 windows per direction (ind = 0, 4, 8)
    # forward model alone: the late query (t = 13) enters in the window starting at 8 -> zeros on frames 0..7 only
    import torch.nn.functional as F
    fr = F.interpolate(frames.float(), (96, 128), mode="bilinear")
    qq = q[0].clone()
    qq[:, 1] *= 0.5
    qq[:, 2] *= 0.75
    tr, vi = CR.cotracker_forward(cot_sd, fr, qq)
    assert (tr[:8, 2] == 0).all() and (tr[8:, 2] != 0).all() and (vi[:8, 2] == 0.5).all()
    short, _ = synthetic_clip(T=5, H=128, W=256, seed=3)
    t5, v5 = CR.CoTrackerTrackerRef(cot_sd, interp_shape=(96, 128)).forward(short[None], q[:, :2])
    assert t5.shape == (1, 5, 2, 2) and torch.isfinite(t5).all()


def test_window_permutation_equivariance(cot_sd):
    """Space attention makes points interact, but the model is equivariant to their order: permuting the queries permutes
    the tracks (to fp32 round-off)."""
    from oracle import cotracker_ref as CR
    import torch.nn.functional as F
    frames, centres = synthetic_clip(T=9, H=96, W=128, seed=5)
    fr = frames.float()
    q = disc_queries(centres, n_pos=5, r=7.0, t=0)
    tr, vi = CR.cotracker_forward(cot_sd, fr, q)
    perm = torch.tensor([3, 0, 4, 1, 2])
    tr2, vi2 = CR.cotracker_forward(cot_sd, fr, q[perm])
    assert (tr[:, perm] - tr2).abs().max() < 1e-3 and (vi[:, perm] - vi2).abs().max() < 1e-4


def test_product_tracker_constructor_matches_reference_yaml():
    """Drop-in: the constructor keywords of configs/model/point_tracker/cotracker.yaml instantiate our class unchanged."""
    import yaml
    from sam_pt_amd.point_tracker import CoTrackerPointTracker, PointTracker
    if os.path.exists(REF_CFG):
        cfg = yaml.safe_load(open(REF_CFG))
    else:   # the GPU box has no /root/reference: the same keys, copied from the yaml (cotracker.yaml:1-11)
        cfg = {"_target_": "sam_pt.point_tracker.cotracker.CoTrackerPointTracker", "checkpoint_path": "x",
               "interp_shape": [384, 512], "visibility_threshold": 0.7, "support_grid_size": 2,
               "support_grid_every_n_frames": 12, "add_debug_visualisations": False}
    assert cfg.pop("_target_").endswith("CoTrackerPointTracker")
    cfg["checkpoint_path"] = None                               # no checkpoint exists here: seeded random weights
    trk = CoTrackerPointTracker(**cfg)
    assert isinstance(trk, PointTracker) and trk.interp_shape == (384, 512) and trk.visibility_threshold == 0.7
    with pytest.raises(Exception):                              # no CPU fallback: fails loudly without a HIP device
        trk(torch.zeros(1, 8, 3, 64, 64, dtype=torch.uint8), torch.zeros(1, 1, 3))


def test_checkpoint_name_selects_the_model_like_build_cotracker(tmp_path):
    """upstream's build_cotracker picks (window, stride) from the checkpoint's FILE NAME and raises on unknown names;
    the three released checkpoints share all weight shapes, so anything else would run silently with the wrong window."""
    from sam_pt_amd.point_tracker import CoTrackerPointTracker, cotracker_model_from_checkpoint_name
    from sam_pt_amd.weights import init_cotracker_state_dict
    assert cotracker_model_from_checkpoint_name("/a/b/cotracker_stride_4_wind_8.pth") == (8, 4)
    assert cotracker_model_from_checkpoint_name("cotracker_stride_4_wind_12.pth") == (12, 4)
    assert cotracker_model_from_checkpoint_name("models/cotracker_ckpts/cotracker_stride_8_wind_16.pth") == (16, 8)
    with pytest.raises(ValueError, match="Unknown model name"):
        cotracker_model_from_checkpoint_name("/x/my_finetune.pth")
    sd = init_cotracker_state_dict(72)
    for name in ("cotracker_stride_4_wind_12.pth", "cotracker_stride_8_wind_16.pth"):
        torch.save({"model": sd}, tmp_path / name)          # loads without a shape error upstream too: same shapes
        with pytest.raises(NotImplementedError, match="window"):
            CoTrackerPointTracker(checkpoint_path=str(tmp_path / name))
    torch.save(sd, tmp_path / "renamed.pth")
    with pytest.raises(ValueError):
        CoTrackerPointTracker(checkpoint_path=str(tmp_path / "renamed.pth"))
    torch.save(sd, tmp_path / "cotracker_stride_4_wind_8.pth")
    assert CoTrackerPointTracker(checkpoint_path=str(tmp_path / "cotracker_stride_4_wind_8.pth")).s == 8
