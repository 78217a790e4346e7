# scratch script of the current gpurun call: persistent GEMM workgroups per launch kind (qkv / proj / fc1 / fc2) beside the faster window chain
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c22; mkdir -p $OUT; cd $R
for w in 30 30/27/30/27 30/27/32/27 32/27/32/27 31/27/31/27 30/24/30/24 30; do
  SAMPT_ENC_WGS=$w timeout 300 python bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline --no-roofline --no-pipelined > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); print("enc wgs $w", d["value"], d.get("timeline"))
except Exception as e: print("bench parse failed $w", e)
PY
done | tee $OUT/enc_wgs_kind.log
