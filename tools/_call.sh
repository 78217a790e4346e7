# scratch script of the current gpurun call: projection + residual + LayerNorm of the image -> token block in one kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c36; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "fused or weights_resident or layernorm" > $OUT/pytest_epi.log 2>&1; tail -6 $OUT/pytest_epi.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_bench_parity.py -x -q -k "dec or sam or golden or predictor or parity or stream or vit" > $OUT/pytest_mod.log 2>&1; tail -3 $OUT/pytest_mod.log | cut -c1-300
for h in 2 1; do SAMPT_GEMM_WRES=$h timeout 400 python bench.py --steps 6 --warmup 2 --no-secondary --no-roofline --no-pipelined --no-cpu-baseline > $OUT/bench_wres$h.json 2> $OUT/bench_wres$h.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_wres$h.json").read().strip().splitlines()[-1]); print("wres $h", d["value"], d.get("timeline"))
except Exception as e: print("bench parse failed", e)
PY
done
