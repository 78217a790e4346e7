# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_v2; mkdir -p $OUT; cd $R
P="--steps 10 --warmup 3 --no-secondary --no-roofline --parity-frames 4"
timeout 400 python bench.py $P --tracker cotracker --neg-points 8 --frames 50 --cotracker-delta-scale 0.001 > $OUT/bench_cfg3_cotracker.log 2>&1; tail -1 $OUT/bench_cfg3_cotracker.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['parity'])"
timeout 200 python tools/gemm_bench.py 8 > $OUT/gemm_microbench.log 2>&1; tail -12 $OUT/gemm_microbench.log
timeout 200 python tools/gemm_bench.py 8 x3 > $OUT/gemm_microbench_x3.log 2>&1; tail -12 $OUT/gemm_microbench_x3.log
timeout 100 python tools/attn_bench.py > $OUT/attn_microbench.log 2>&1; timeout 100 python tools/attn_bench.py x3 >> $OUT/attn_microbench.log 2>&1; cat $OUT/attn_microbench.log
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-secondary --no-roofline"
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o vith -- python $R/bench.py $Q --steps 5 --warmup 2 > $OUT/rocprof_f16.log 2>&1
cd $R; python tools/rocprof_summary.py $(find $OUT/prof -name "*.db" | head -1) 312 > $OUT/vith_kernel_stats.txt 2>&1; head -24 $OUT/vith_kernel_stats.txt; rm -rf $OUT/prof
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o vith -- python $R/bench.py $Q --steps 3 --warmup 1 --precision f16x3 > $OUT/rocprof_x3.log 2>&1
cd $R; python tools/rocprof_summary.py $(find $OUT/prof -name "*.db" | head -1) 216 > $OUT/vith_f16x3_kernel_stats.txt 2>&1; head -14 $OUT/vith_f16x3_kernel_stats.txt; rm -rf $OUT/prof
