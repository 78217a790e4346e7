# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c13; mkdir -p $OUT; cd $R
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 300 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "prefetch or stream_of_clips or dropin" > $OUT/pytest_prefetch.log 2>&1; tail -1 $OUT/pytest_prefetch.log
