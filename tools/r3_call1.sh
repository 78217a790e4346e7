set -x
mkdir -p gpurun_out/r3a
cd /root/repo
timeout 150 python tools/gemm_bench.py 8 > gpurun_out/r3a/p8_default.log 2>&1; echo rc=$?
tail -30 gpurun_out/r3a/p8_default.log
SAMPT_GEMM_P8=0 timeout 100 python tools/gemm_bench.py 8 nocheck > gpurun_out/r3a/old.log 2>&1; echo rc=$?
SAMPT_GEMM_STAGGER=0 timeout 100 python tools/gemm_bench.py 8 nocheck > gpurun_out/r3a/p8_nostagger.log 2>&1; echo rc=$?
SAMPT_GEMM_R=8 timeout 100 python tools/gemm_bench.py 8 nocheck > gpurun_out/r3a/p8_r8.log 2>&1; echo rc=$?
SAMPT_GEMM_R=2 timeout 100 python tools/gemm_bench.py 8 nocheck > gpurun_out/r3a/p8_r2.log 2>&1; echo rc=$?
SAMPT_GEMM_GELU_FAST=0 timeout 100 python tools/gemm_bench.py 8 nocheck > gpurun_out/r3a/p8_erff.log 2>&1; echo rc=$?
timeout 100 python tools/gemm_bench.py 24 nocheck > gpurun_out/r3a/p8_b24.log 2>&1; echo rc=$?
timeout 100 python tools/gemm_bench.py 8 nocheck zeros > gpurun_out/r3a/p8_zeros.log 2>&1; echo rc=$?
for f in old p8_nostagger p8_r8 p8_r2 p8_erff p8_b24 p8_zeros; do echo "== $f"; grep -v "^check" gpurun_out/r3a/$f.log | tail -12; done
