set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v15; mkdir -p $OUT; cd $R
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-secondary --no-roofline"
for b in 8 12 24 8 12; do timeout 200 python bench.py $Q --encode-batch $b > $OUT/bench_eb$b.log 2>&1; echo "encode-batch $b: $(tail -1 $OUT/bench_eb$b.log | cut -c80-150)"; done
cd /tmp; export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace -d $OUT/prof -o vith -- python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline > $OUT/rocprof.log 2>&1
DB=$(ls $OUT/prof/*/vith_results.db $OUT/prof/vith_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py "$DB" 144 > $OUT/vith_kernel_stats.txt 2>&1
rm -rf $OUT/prof
head -30 $OUT/vith_kernel_stats.txt | cut -c1-140
