set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v7; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_modules.py tests/test_gpu_bench_parity.py -x -q -m gpu > $OUT/pytest_vit.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_vit.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_skip.log 2>&1
SAMPT_VIT_SKIP_DEAD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_noskip.log 2>&1
timeout 200 python tools/attn_bench.py > $OUT/attn_default.log 2>&1
SAMPT_FLASH_NW7=1 timeout 200 python tools/attn_bench.py > $OUT/attn_nw7.log 2>&1
SAMPT_FLASH_NW7=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_nw7.log 2>&1
tail -4 $OUT/pytest_vit.log; for f in bench_skip bench_noskip bench_nw7; do tail -1 $OUT/$f.log | cut -c1-160; done; tail -5 $OUT/attn_default.log; tail -5 $OUT/attn_nw7.log
