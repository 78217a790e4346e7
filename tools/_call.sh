set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c33; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_modules.py -x -q > $OUT/pytest_mod.log 2>&1; tail -3 $OUT/pytest_mod.log | cut -c1-300
timeout 300 python tools/stray_ops.py > $OUT/stray_ops.log 2>&1; grep -v "amdgpu.ids\|Warn\|warn" $OUT/stray_ops.log | tail -8 | cut -c1-250
