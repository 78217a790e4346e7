# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c14; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests/test_gpu_bench_parity.py -q -m gpu -s -k "cfg3 or cfg5" > $OUT/pytest_cfg3.log 2>&1; grep -E "config parity|passed|failed" $OUT/pytest_cfg3.log | cut -c1-900
