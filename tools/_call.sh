# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r5_*)
# r5_final: the round's validation — full GPU suite, smoke(), the default bench line with live oracle / cpu_baseline / roofline /
# secondaries (what the driver runs), and the fp32-grade mode as a full 20-step line with its own roofline block
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_final; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 900 python bench.py > $OUT/bench_default.log 2>&1
timeout 300 python bench.py --precision f16x3 --no-cpu-baseline --no-secondary > $OUT/bench_f16x3.log 2>&1
tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log | cut -c1-300; tail -1 $OUT/bench_default.log | cut -c1-700; tail -1 $OUT/bench_f16x3.log | cut -c1-500
