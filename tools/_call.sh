# scratch script of the current gpurun call: re-check after moving the bias-correction host math into pack.vit_bias_correction
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_c11; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests/test_gpu_modules.py -q -s -k "bias_correction or vit_b_encoder or dead_row or prefetch" > $OUT/pytest_bias.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 10 --warmup 3 > $OUT/bench_quick.log 2>&1
tail -3 $OUT/pytest_bias.log; tail -1 $OUT/bench_quick.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_pipelined'], d['parity']['mask_iou_min'], d['parity']['pass'])"
