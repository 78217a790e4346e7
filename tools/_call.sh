# scratch script of the current gpurun call: InstanceNorm statistics from the halo convolution's epilogue
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c20; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "conv or instance" > $OUT/pytest_conv.log 2>&1; tail -5 $OUT/pytest_conv.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_modules.py -q -k "fnet or golden or pips" > $OUT/pytest_fnet.log 2>&1; tail -3 $OUT/pytest_fnet.log | cut -c1-300
for h in 3 1; do SAMPT_CONV_HALO=$h timeout 100 python tools/tracker_bench.py 2>&1 | grep "tracker encoder"; done
for h in 3 1; do SAMPT_CONV_HALO=$h timeout 400 python bench.py --steps 4 --warmup 2 --no-secondary --no-roofline --no-pipelined > $OUT/bench_halo$h.json 2> $OUT/bench_halo$h.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_halo$h.json").read().strip().splitlines()[-1]); print("halo $h", d["value"], d.get("timeline"), d.get("parity"))
except Exception as e: print("bench parse failed", e)
PY
done
