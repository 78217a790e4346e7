# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c8; mkdir -p $OUT; cd $R
for w in 28 30 32 28 26; do
  SAMPT_ENC_WGS=$w timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_wgs$w.log 2>&1; tail -1 $OUT/bench_wgs$w.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wgs $w', d['value'], d['value_per_forward'])"
done
for w in 28 32; do
  SAMPT_ENC_WGS=$w timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --precision f16x3 > $OUT/bench_x3_wgs$w.log 2>&1; tail -1 $OUT/bench_x3_wgs$w.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('x3 wgs $w', d['value'], d['value_per_forward'])"
done
