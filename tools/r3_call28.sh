set -x
mkdir -p gpurun_out/r3x
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/r3x/pytest_gpu.log 2>&1; tail -18 gpurun_out/r3x/pytest_gpu.log
