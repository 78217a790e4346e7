# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c10; mkdir -p $OUT; cd $R
for a in 1 0; do
  SAMPT_FLASH_ALL=$a timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "vit_flash_attention or window_attention" > $OUT/pytest_flash_all$a.log 2>&1; tail -1 $OUT/pytest_flash_all$a.log
  SAMPT_FLASH_ALL=$a timeout 100 python tools/attn_bench.py > $OUT/attn_all$a.log 2>&1; tail -2 $OUT/attn_all$a.log
done
SAMPT_FLASH_ALL=1 timeout 300 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "vit_ or dead_row" > $OUT/pytest_vit_all1.log 2>&1; tail -1 $OUT/pytest_vit_all1.log
for a in 1 0 1 0; do
  SAMPT_FLASH_ALL=$a timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_all$a.log 2>&1; tail -1 $OUT/bench_all$a.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('all $a', d['value'], d['value_per_forward'], d['parity']['mask_iou_min'])"
done
