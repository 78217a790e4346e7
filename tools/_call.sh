set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v18; mkdir -p $OUT; cd $R
timeout 200 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "both_arithmetics" > $OUT/pytest_f32dec.log 2>&1; echo "rc=$?" >> $OUT/pytest_f32dec.log
tail -15 $OUT/pytest_f32dec.log
