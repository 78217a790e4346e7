# scratch script of the current gpurun call: first run of the halo-tiled 3 x 3 convolution (csrc/conv_halo_x3.hip): conv / fnet tests,
# micro-benchmark against the implicit-GEMM kernel, tracker encoder time and quick bench both ways
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c13; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "conv" > $OUT/pytest_conv.log 2>&1; tail -15 $OUT/pytest_conv.log | cut -c1-300
timeout 200 python tools/conv_halo_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_halo_bench.log
timeout 600 python -m pytest tests/test_gpu_modules.py -q -k "fnet or tracker_vs or golden or update_window" > $OUT/pytest_fnet.log 2>&1; tail -3 $OUT/pytest_fnet.log
for on in 0 1; do
  SAMPT_CONV_HALO=$on timeout 100 python tools/tracker_bench.py 2>&1 | grep "tracker encoder" | sed "s/^/halo=$on: /"
  SAMPT_CONV_HALO=$on timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 10 --warmup 3 > $OUT/bench_halo$on.log 2>&1
  tail -1 $OUT/bench_halo$on.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('halo=$on', d['value'], d.get('timeline'), d['parity']['mask_iou_min'], d['parity']['pass'], d['parity']['traj_max_abs_px'])"
done
