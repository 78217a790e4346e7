# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c24; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ATTN_BENCH_LIB=$R/tools/_ab/libsampt_hip_head.so timeout 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/old -- python $R/tools/attn_bench.py > $OUT/old.log 2>&1
timeout 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/new -- python $R/tools/attn_bench.py > $OUT/new.log 2>&1
cd $R
echo "# previous commit's kernels (3-D grid, dispatch order)" > $OUT/attn_fetch_size.txt; python tools/pmc_summary.py $OUT/old flash >> $OUT/attn_fetch_size.txt
echo "# this build (1-D grid, windows XCD-aware)" >> $OUT/attn_fetch_size.txt; python tools/pmc_summary.py $OUT/new flash >> $OUT/attn_fetch_size.txt
rm -rf $OUT/old $OUT/new; cat $OUT/attn_fetch_size.txt; tail -2 $OUT/new.log
