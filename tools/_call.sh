set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c42; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "conv or instnorm" > $OUT/pytest_conv.log 2>&1; tail -3 $OUT/pytest_conv.log | cut -c1-300
timeout 200 python tools/conv_halo_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_halo_bench.log | cut -c1-300
timeout 400 python tools/probes/conv_stats_determinism.py 100 1 2>&1 | grep -v "amdgpu.ids" | tail -12 | cut -c1-200
for h in 8 1; do SAMPT_CONV_HALO=$h timeout 100 python tools/tracker_bench.py 2>&1 | grep "tracker encoder" | sed "s/^/halo $h: /"; done
