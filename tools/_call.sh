# scratch script of the current gpurun call: weights-resident split-fp16 GEMM for the decoder's image-side projections
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c18; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "weights_resident" > $OUT/pytest_wres.log 2>&1; tail -5 $OUT/pytest_wres.log | cut -c1-300
timeout 200 python tools/gemm_wres_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_wres_bench.log
timeout 900 python -m pytest tests/test_gpu_modules.py -q -x -k "dec or sam or golden or predictor" > $OUT/pytest_dec.log 2>&1; tail -3 $OUT/pytest_dec.log | cut -c1-300
for h in 0 1; do SAMPT_GEMM_WRES=$h timeout 400 python bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-roofline --no-pipelined > $OUT/bench_wres$h.json 2> $OUT/bench_wres$h.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_wres$h.json").read().strip().splitlines()[-1]); print("wres $h", d["value"], d.get("timeline"))
except Exception as e: print("bench parse failed", e)
PY
done
