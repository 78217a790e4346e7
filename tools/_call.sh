set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c11; mkdir -p $OUT; cd $R
timeout 60 tools/probes/cu_mask_probe > $OUT/cu_mask_probe.log 2>&1; cat $OUT/cu_mask_probe.log
timeout 300 python -m pytest tests/test_gpu_modules.py -q -k "fnet or tracker_vs or golden" > $OUT/pytest_fnet.log 2>&1; tail -2 $OUT/pytest_fnet.log
timeout 100 python tools/tracker_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/tracker_bench.log
