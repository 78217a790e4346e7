set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v9; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_parity.py -x -q -m gpu > $OUT/pytest_kernels2.log 2>&1; echo "rc=$?" >> $OUT/pytest_kernels2.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_vith.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-secondary --no-roofline --hq --tracker cotracker --points 16 --objects 5 --square 1024 --frames 24 > $OUT/bench_cfg5_hq_cotracker.log 2>&1
timeout 100 python tools/forward_timeline.py > $OUT/forward_timeline.log 2>&1
tail -3 $OUT/pytest_kernels2.log; tail -1 $OUT/bench_vith.log | cut -c1-400; tail -1 $OUT/bench_cfg5_hq_cotracker.log | cut -c1-200; tail -2 $OUT/forward_timeline.log
