# scratch script of the current gpurun call: kernel trace of the tracker encoder after the halo convolution
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c16; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $OUT/prof -o trk -- python $R/tools/tracker_bench.py > $OUT/rocprof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python $R/tools/rocprof_by_grid.py "$DB" "" 12 > $OUT/tracker_by_grid.txt 2>&1
rm -rf $OUT/prof
grep "tracker" $OUT/rocprof.log
head -45 $OUT/tracker_by_grid.txt | cut -c1-150
