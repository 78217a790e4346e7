set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v21; mkdir -p $OUT; cd $R
timeout 100 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 100 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "fused or ragged or multi or negative" > $OUT/pytest_fused.log 2>&1; tail -2 $OUT/pytest_fused.log
timeout 100 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --objects 3 > $OUT/bench_3obj.log 2>&1; echo "3obj: $(tail -1 $OUT/bench_3obj.log | cut -c80-150)"
