# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c12; mkdir -p $OUT; cd $R
Q="--steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['value_per_forward'], d.get('parity',{}).get('mask_iou_min'))" 2>&1 | tail -1; }
timeout 300 python bench.py $Q --hq --precision f16x3 > $OUT/bench_hq_x3.log 2>&1; show $OUT/bench_hq_x3.log hq_x3
timeout 300 python bench.py $Q --model vit_l --precision f16x3 > $OUT/bench_vitl_x3.log 2>&1; show $OUT/bench_vitl_x3.log vitl_x3
timeout 300 python bench.py $Q --model vit_l > $OUT/bench_vitl_f16.log 2>&1; show $OUT/bench_vitl_f16.log vitl_f16
timeout 300 python bench.py $Q --model vit_b --precision f16x3 > $OUT/bench_vitb_x3.log 2>&1; show $OUT/bench_vitb_x3.log vitb_x3
timeout 300 python bench.py $Q --native-480p --precision f16x3 > $OUT/bench_native_x3.log 2>&1; show $OUT/bench_native_x3.log native_x3
timeout 300 python - > $OUT/vitl_x3_vs_f32.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from sam_pt_amd.sam_predictor import SamHip, SamPredictor
from sam_pt_amd.synth import bench_clip
dev = torch.device("cuda:0")
frames, _ = bench_clip(T=2)
embs = {}
for prec in ("f32", "f16x3", "f16"):
    pred = SamPredictor(SamHip("vit_l", precision=prec, seed=72, max_batch=2).to(dev))
    embs[prec] = pred.encode_frames(frames.to(dev)).float().cpu()
    del pred; torch.cuda.empty_cache()
ref = embs["f32"]
for prec in ("f16x3", "f16"):
    print("vit_l", prec, "vs f32: rel err", float((embs[prec] - ref).abs().max() / ref.abs().max()))
PY
cat $OUT/vitl_x3_vs_f32.log | tail -3
