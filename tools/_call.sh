# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c6; mkdir -p $OUT; cd $R
for nb in 1 2; do
  SAMPT_X3_NBUF=$nb timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "flash_attention_x3 or window_attention" > $OUT/pytest_x3_nbuf$nb.log 2>&1; tail -2 $OUT/pytest_x3_nbuf$nb.log
  SAMPT_X3_NBUF=$nb timeout 100 python tools/attn_bench.py x3 > $OUT/attn_x3_nbuf$nb.log 2>&1; tail -2 $OUT/attn_x3_nbuf$nb.log
done
for nb in 1 2 1 2; do
  SAMPT_X3_NBUF=$nb timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --precision f16x3 > $OUT/bench_x3_nbuf$nb.log 2>&1; tail -1 $OUT/bench_x3_nbuf$nb.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nbuf $nb', d['value'], d['value_per_forward'], d['parity']['mask_iou_min'], d['parity']['pass'])"
done
timeout 600 python -m pytest tests/test_gpu_bench_parity.py -q -m gpu -s -k "cfg5" > $OUT/pytest_cfg5.log 2>&1; grep "parity\]\|passed\|failed" $OUT/pytest_cfg5.log | cut -c1-600
