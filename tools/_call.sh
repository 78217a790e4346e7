# scratch script of the current gpurun call: the GPU suite exactly as a fresh clone sees it (no oracle cache: every oracle result live)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_final3; mkdir -p $OUT; cd $R
rm -rf tests/golden/oracle_cache          # (the box's copy of the tree is scratch)
ls tests/golden/oracle_cache > $OUT/cache_state.log 2>&1
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu_live_oracle.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_live_oracle.log
cat $OUT/cache_state.log; tail -8 $OUT/pytest_gpu_live_oracle.log
