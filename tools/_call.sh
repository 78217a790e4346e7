# scratch script of the current gpurun call: LayerNorm2d + GELU and the mask dot product in the weights-resident GEMMs' epilogues
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c35; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "fused_upscaling or weights_resident or layernorm" > $OUT/pytest_epi.log 2>&1; tail -6 $OUT/pytest_epi.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_bench_parity.py -x -q -k "dec or sam or golden or predictor or parity or stream" > $OUT/pytest_mod.log 2>&1; tail -3 $OUT/pytest_mod.log | cut -c1-300
for h in 2 1; do SAMPT_GEMM_WRES=$h timeout 400 python bench.py --steps 4 --warmup 2 --no-secondary --no-roofline --no-pipelined > $OUT/bench_wres$h.json 2> $OUT/bench_wres$h.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_wres$h.json").read().strip().splitlines()[-1]); print("wres $h", d["value"], d.get("timeline"), "parity", d["parity"]["pass"], d["parity"]["mask_iou_min"], d["parity"]["logit_max_abs"])
except Exception as e: print("bench parse failed", e)
PY
done
