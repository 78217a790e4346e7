# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c21; mkdir -p $OUT; cd $R
timeout 420 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
Q="--no-cpu-baseline --no-secondary"
timeout 150 python bench.py $Q > $OUT/bench_vith.log 2>&1; tail -1 $OUT/bench_vith.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace -d $OUT/prof -o vith -- python $R/bench.py $Q --no-roofline --steps 5 --warmup 2 > $OUT/rocprof.log 2>&1
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 288 > $OUT/vith_kernel_stats.txt 2>&1
rm -rf $OUT/prof
grep -E "flash|total kernel" $OUT/vith_kernel_stats.txt | cut -c1-160
