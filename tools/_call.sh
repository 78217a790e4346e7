set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v19; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $OUT/prof -o cot -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --tracker cotracker --points 8 --neg-points 8 --frames 50 > $OUT/rocprof_cfg3.log 2>&1
DB=$(ls $OUT/prof/*/cot_results.db $OUT/prof/cot_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py "$DB" 150 > $OUT/cfg3_cotracker_kernel_stats.txt 2>&1
rm -rf $OUT/prof
head -26 $OUT/cfg3_cotracker_kernel_stats.txt | cut -c1-150
