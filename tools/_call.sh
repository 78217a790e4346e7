# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r5_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_c3; mkdir -p $OUT; cd $R
# 1. the 8-phase GEMM with its LDS-DMA issued in the read segments (SAMPT_GEMM_SCHED=0, the new default) vs the round-3 schedule (1):
#    correctness cases of tools/gemm_bench.py first, then the ViT-H shapes, both arithmetics
( export SAMPT_GEMM_SCHED=0; timeout 300 python tools/gemm_bench.py 8 > $OUT/gemm_sched0.log 2>&1 )
( export SAMPT_GEMM_SCHED=1; timeout 300 python tools/gemm_bench.py 8 nocheck > $OUT/gemm_sched1.log 2>&1 )
( export SAMPT_GEMM_SCHED=0; timeout 300 python tools/gemm_bench.py 8 x3 > $OUT/gemm_x3_sched0.log 2>&1 )
( export SAMPT_GEMM_SCHED=1; timeout 300 python tools/gemm_bench.py 8 x3 > $OUT/gemm_x3_sched1.log 2>&1 )
# 2. phase groups (staggered starts) on the better... both schedules
( export SAMPT_GEMM_SCHED=0 SAMPT_GEMM_STAGGER=2; timeout 300 python tools/gemm_bench.py 8 nocheck > $OUT/gemm_sched0_stagger2.log 2>&1 )
( export SAMPT_GEMM_SCHED=0 SAMPT_GEMM_STAGGER=3; timeout 300 python tools/gemm_bench.py 8 nocheck > $OUT/gemm_sched0_stagger3.log 2>&1 )
# 3. kernel + encoder tests under the new default
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q -x -k "gemm or vit" > $OUT/pytest_gemm_vit.log 2>&1
# 4. in situ
Q="--no-cpu-baseline --no-secondary --steps 20 --warmup 5"
run() { echo "== $1" >> $OUT/bench_ab.log; ( export $1; timeout 200 python bench.py $Q $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('blocking', d['value'], 'pipelined', d['value_pipelined'], 'parity', d.get('parity',{}).get('pass'), d.get('parity',{}).get('mask_iou_min'), 'gemm in situ', r.get('achieved'), r.get('frac'), 'isolated', r.get('isolated_achieved'))" ) >> $OUT/bench_ab.log 2>&1; }
run "SAMPT_GEMM_SCHED=1" ""
run "SAMPT_GEMM_SCHED=0" ""
run "SAMPT_GEMM_SCHED=0 SAMPT_GEMM_STAGGER=2" ""
run "SAMPT_GEMM_SCHED=1" ""
run "SAMPT_GEMM_SCHED=0" ""
run "SAMPT_GEMM_SCHED=1" "--precision f16x3 --steps 8 --warmup 2"
run "SAMPT_GEMM_SCHED=0" "--precision f16x3 --steps 8 --warmup 2"
for f in gemm_sched0 gemm_sched1 gemm_x3_sched0 gemm_x3_sched1 gemm_sched0_stagger2 gemm_sched0_stagger3; do echo "---- $f"; grep -E "ALL CHECKS|FAIL|mix|Error|error" $OUT/$f.log | head -5; done
tail -3 $OUT/pytest_gemm_vit.log; cat $OUT/bench_ab.log
