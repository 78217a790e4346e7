# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c9; mkdir -p $OUT; cd $R
for sb in 1 0; do
  SAMPT_CONV_SB=$sb timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv_f16x3" > $OUT/pytest_conv_sb$sb.log 2>&1; tail -1 $OUT/pytest_conv_sb$sb.log
  SAMPT_CONV_SB=$sb timeout 200 python tools/stage_times.py > $OUT/stage_times_sb$sb.log 2>&1; tail -4 $OUT/stage_times_sb$sb.log
done
SAMPT_CONV_SB=1 timeout 400 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "predict_torch or hq_ or decoder or fnet or vit_b_encoder" > $OUT/pytest_modules_sb1.log 2>&1; tail -1 $OUT/pytest_modules_sb1.log
for sb in 1 0 1 0; do
  SAMPT_CONV_SB=$sb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_sb$sb.log 2>&1; tail -1 $OUT/bench_sb$sb.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sb $sb', d['value'], d['value_per_forward'], d['parity']['mask_iou_min'])"
done
for sb in 1 0; do
  SAMPT_CONV_SB=$sb timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --hq --tracker cotracker --square 1024 --points 16 --objects 5 > $OUT/bench_cfg5_sb$sb.log 2>&1; tail -1 $OUT/bench_cfg5_sb$sb.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 sb $sb', d['value'], d['value_per_forward'])"
done
