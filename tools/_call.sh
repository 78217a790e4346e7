set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v18; mkdir -p $OUT; cd $R
SAMPT_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "opt_in" > $OUT/pytest_optin.log 2>&1; echo "rc=$?" >> $OUT/pytest_optin.log
SAMPT_DEC_F16X3=0 timeout 200 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "hq or predict_torch or large_prompt" > $OUT/pytest_f32dec_more.log 2>&1; echo "rc=$?" >> $OUT/pytest_f32dec_more.log
tail -3 $OUT/pytest_optin.log; tail -3 $OUT/pytest_f32dec_more.log
