# scratch script of the current gpurun call: rel-pos tables as host-packed MFMA operand images in the fp16 attention kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c43; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_kernels.py -x -q -k "vit or attention" > $OUT/pytest_vit.log 2>&1; tail -3 $OUT/pytest_vit.log | cut -c1-300
for h in 0 1 0 1; do SAMPT_ATTN_REL_OPS=$h timeout 400 python bench.py --steps 6 --warmup 2 --no-secondary --no-roofline --no-pipelined --no-cpu-baseline > $OUT/bench_ops$h.json 2> $OUT/bench_ops$h.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_ops$h.json").read().strip().splitlines()[-1]); print("rel ops $h", d["value"], d.get("timeline"))
except Exception as e: print("bench parse failed", e)
PY
done | tee $OUT/bench_ab.log
