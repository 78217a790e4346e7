set -x
mkdir -p gpurun_out/r3q2
cd /root/repo
timeout 500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_bench_parity.py --ignore tests/test_gpu_bench_parity.py > gpurun_out/r3q2/pytest_gpu_fast.log 2>&1; tail -4 gpurun_out/r3q2/pytest_gpu_fast.log
timeout 200 python -m pytest tests/test_gpu_bench_parity.py -q -m gpu -k "cfg4 or vit_b" > gpurun_out/r3q2/pytest_parity_sub.log 2>&1; tail -3 gpurun_out/r3q2/pytest_parity_sub.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r3q2/bench.log 2>&1; echo "bench: $(tail -1 gpurun_out/r3q2/bench.log | grep -o '"value": [0-9.]*')"
