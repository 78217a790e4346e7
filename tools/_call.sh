set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v1; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "attention or resize_logits or vos_index" > $OUT/pytest_kernels.log 2>&1
timeout 420 python -m pytest tests/test_gpu_cotracker.py -q -m gpu -s > $OUT/pytest_cotracker.log 2>&1
timeout 300 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_modules.py -q -m gpu -s -k "bench_clip or prompt_size or vit_b_encoder or vit_test_encoder" > $OUT/pytest_parity.log 2>&1
timeout 120 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_vith.log 2>&1
tail -3 $OUT/pytest_kernels.log; tail -3 $OUT/pytest_cotracker.log; tail -3 $OUT/pytest_parity.log; tail -1 $OUT/bench_vith.log | cut -c1-200
