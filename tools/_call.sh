set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v6; mkdir -p $OUT; cd $R
for v in 1 9 4 7 3; do
  SAMPT_GEMM_BN160=0 SAMPT_GEMM_VARIANT=$v timeout 100 python tools/gemm_bench.py 8 > $OUT/gemm_v$v.log 2>&1
done
timeout 100 python tools/gemm_bench.py 8 > $OUT/gemm_bn160.log 2>&1
for v in 1 9 4 7 3 bn160; do echo "== variant $v"; grep "M= 32768" $OUT/gemm_v$v.log $OUT/gemm_$v.log 2>/dev/null | cut -d: -f2- | cut -c1-100; done
timeout 90 python tools/stage_times.py > $OUT/stage_times.log 2>&1; tail -1 $OUT/stage_times.log
