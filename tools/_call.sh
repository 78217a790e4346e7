# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c7; mkdir -p $OUT; cd $R
for nb in 1 2; do
  SAMPT_FLASH_NBUF=$nb timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "vit_flash_attention or window_attention" > $OUT/pytest_flash_nbuf$nb.log 2>&1; tail -1 $OUT/pytest_flash_nbuf$nb.log
  SAMPT_FLASH_NBUF=$nb timeout 100 python tools/attn_bench.py > $OUT/attn_nbuf$nb.log 2>&1; tail -2 $OUT/attn_nbuf$nb.log
done
for nb in 1 2 1 2; do
  SAMPT_FLASH_NBUF=$nb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_nbuf${nb}.log 2>&1; tail -1 $OUT/bench_nbuf${nb}.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nbuf $nb', d['value'], d['value_per_forward'], d['parity']['mask_iou_min'])"
done
