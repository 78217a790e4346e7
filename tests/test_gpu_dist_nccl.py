"""RCCL readiness on the one GPU a test box has.  No 8-GPU node is available to the build,
so the N > 1 data path is covered by the 2-process gloo tests on the CPU (tests/test_cpu_host.py) — which cannot tell whether
the ``nccl`` backend (RCCL on ROCm) initialises, accepts the tensors this package hands it and runs the two collectives of the
multi-GPU modes on device memory.  These tests do exactly that at world size 1 through ``torch.distributed.run``:

* ``dist.sharded_forward(..., shard_fnet=True, force_collectives=True)``: ``FnetShard.exchange`` = one
  ``all_gather_into_tensor`` per pyramid level on the tracker encoder's own buffers, then ``gather_masks`` = one ``gather`` of the
  uint8 index masks — the result must equal the plain forward's;
* ``bench.py --gpus 1`` launched the way the driver launches N > 1 (RANK / WORLD_SIZE / MASTER_* from the environment).
"""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(args, timeout=600):
    port = str(_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", port] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_rccl_collectives_of_the_sharded_modes_at_world_size_one():
    code = r"""
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from sam_pt_amd.dist import FnetShard, gather_masks, index_masks, sharded_forward
from sam_pt_amd.point_tracker import PipsPointTracker
from sam_pt_amd.sam_predictor import SamHip, SamPredictor
from sam_pt_amd.sam_pt import SamPt
from sam_pt_amd.synth import disc_queries, synthetic_clip
from sam_pt_amd.weights import SAM_CONFIGS, init_pips_state_dict, init_sam_state_dict
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
cfg = SAM_CONFIGS["vit_test"]
sd, psd = init_sam_state_dict(cfg, 72), init_pips_state_dict(72)
frames, centres = synthetic_clip(T=6, H=128, W=256, seed=72)
q = disc_queries(centres, n_pos=3, r=9.0)
video = {"image": [f.to(dev) for f in frames], "target_hw": (128, 256), "query_points": q[None]}
def model():
    return SamPt(PipsPointTracker(state_dict=psd), SamPredictor(SamHip(config=cfg, state_dict=sd, precision="f32").to(dev)),
                 sam_iou_threshold=-1e9, positive_points_per_mask=3, negative_points_per_mask=0,
                 iterative_refinement_iterations=1).eval()
ref = model()(video)
want = index_masks(torch.stack(ref["logits"], dim=0))
full, own = sharded_forward(model(), video, batch=2, shard_fnet=True, force_collectives=True)
torch.cuda.synchronize()
fs = own["fnet_shard"]
assert isinstance(fs, FnetShard) and not fs.emulate                      # the all_gather ran (RCCL, device buffers)
assert torch.equal(full, want) and int(want.sum()) > 0
assert torch.equal(own["trajectories"].cpu(), ref["trajectories"].cpu())
# the mask gather alone, padded shape, uint8 on the device
g = gather_masks(want, 9, force=True)
assert g.shape == (1, 9) + tuple(want.shape[1:]) and torch.equal(g[0, :6], want) and int(g[0, 6:].sum()) == 0
dist.barrier(); dist.destroy_process_group()
print("RCCL_WORLD1_OK")
""" % ROOT
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
        path = f.name
    try:
        r = _torchrun([path])
    finally:
        os.unlink(path)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RCCL_WORLD1_OK" in r.stdout


def test_bench_under_torch_distributed_run():
    """bench.py launched as the driver launches it for N > 1 (rendezvous from the environment), one rank: one JSON line with the
    contract's fields, `clips_encoded == steps`."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--model", "vit_b", "--frames", "8",
                   "--no-cpu-baseline", "--no-secondary", "--no-roofline"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["clips_encoded"] == 2 and d["value"] > 0 and d["value_pipelined"] > 0
