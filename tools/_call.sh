set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c38; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q -k "shi_tomasi or kmedoid or erode or query" > $OUT/pytest_qp.log 2>&1; tail -4 $OUT/pytest_qp.log | cut -c1-300
