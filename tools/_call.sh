# scratch script of the current gpurun call: the round's validation on the final tree — full GPU suite, smoke(), the default bench line
# (live oracle, cpu_baseline, roofline, secondaries, f16x3 side line), kernel trace of the bench command
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_final; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log | cut -c1-200
timeout 2400 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-400
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o vith -- python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 5 --warmup 2 > $OUT/rocprof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" 288 > $OUT/vith_kernel_stats.txt 2>&1
python $R/tools/rocprof_by_grid.py "$DB" "" 12 > $OUT/vith_kernels_by_grid.txt 2>&1
rm -rf $OUT/prof
head -6 $OUT/vith_kernel_stats.txt | cut -c1-160
