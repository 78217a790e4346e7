# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r5_*)
# r5_final2: re-validation of the committed tree after the last edits (emulation test for both shard modes, thin-GEMM knob, docs):
# full GPU suite, smoke(), the default bench line exactly as the driver runs it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_final2; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>&1
tail -4 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log | cut -c1-200; tail -1 $OUT/bench_default.log | cut -c1-500
