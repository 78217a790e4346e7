import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
args = bench.parse()
dev = torch.device("cuda:0")
from sam_pt_amd.synth import bench_clip
frames, qp = bench_clip(T=args.frames, seed=72, n_pos=args.points, n_objects=args.objects)
model = bench.build_model(args, dev)
video = {"image": [f for f in frames.to(dev)], "target_hw": tuple(frames.shape[-2:]), "query_points": qp}
for _ in range(2): model(video)
torch.cuda.synchronize()
images = frames.to(dev)
pred = model.sam_predictor
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
feats, t_enc = T(lambda: pred.encode_frames(images))
(tr, vi), t_trk = T(lambda: model._track_points(images, qp))
qm, t_qm = T(lambda: model.extract_query_masks(images, qp, feats))
(_, logits, spf), t_dec = T(lambda: model._apply_sam_to_trajectories(images, tr, vi, feats))
from sam_pt_amd.dist import index_masks
_, t_idx = T(lambda: index_masks(logits))
_, t_all = T(lambda: model(video))
print(f"encode {t_enc:.1f} ms | tracker {t_trk:.1f} | query masks {t_qm:.1f} | decode {t_dec:.1f} | index {t_idx:.1f} | forward total {t_all:.1f} (sum serial {t_enc+t_trk+t_qm+t_dec:.1f})")
