# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c23; mkdir -p $OUT; cd $R
timeout 100 python tools/gemm_bench.py 8 2>&1 | grep -v amdgpu.ids > $OUT/gemm_atomic.log
SAMPT_P8_ATOMIC=0 timeout 80 python tools/gemm_bench.py 8 nocheck 2>&1 | grep -v amdgpu.ids > $OUT/gemm_noatomic.log
grep -E "inplace=1|ALL CHECKS|FAIL" $OUT/gemm_atomic.log | cut -c1-160; grep -E "proj|fc2|mix" $OUT/gemm_atomic.log; echo ---; grep -E "proj|fc2|mix" $OUT/gemm_noatomic.log
