# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c26; mkdir -p $OUT; cd $R
timeout 150 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 150 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "flash_attention or window_attention or gemm" 2>&1 | tail -1 | tee $OUT/pytest_kernels.log
