# scratch script of the current gpurun call: split-fp16 stem convolution with fused InstanceNorm statistics
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c23; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "stem or instnorm" > $OUT/pytest_stem.log 2>&1; tail -5 $OUT/pytest_stem.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_modules.py -q -k "fnet or golden or pips" > $OUT/pytest_fnet.log 2>&1; tail -3 $OUT/pytest_fnet.log | cut -c1-300
for h in 0 3 1; do SAMPT_CONV_HALO=$h timeout 100 python tools/tracker_bench.py 2>&1 | grep "tracker encoder" | sed "s/^/halo $h: /"; done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o trk -- python $R/tools/tracker_bench.py > $OUT/rocprof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python $R/tools/rocprof_by_grid.py "$DB" "" 5 > $OUT/tracker_by_grid.txt 2>&1
rm -rf $OUT/prof
grep -v "pips_mix\|k_pips\|thin\|skinny" $OUT/tracker_by_grid.txt | head -30 | cut -c1-150
