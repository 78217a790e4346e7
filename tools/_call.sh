# scratch script of the current gpurun call: halo convolution, 64-channel variant with one halo buffer and two workgroups per CU
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c14; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "conv" > $OUT/pytest_conv.log 2>&1; tail -2 $OUT/pytest_conv.log | cut -c1-300
timeout 200 python tools/conv_halo_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_halo_bench.log
timeout 300 python -m pytest tests/test_gpu_modules.py -q -k "fnet or golden" > $OUT/pytest_fnet.log 2>&1; tail -2 $OUT/pytest_fnet.log
timeout 100 python tools/tracker_bench.py 2>&1 | grep "tracker encoder"
