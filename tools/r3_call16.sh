set -x
mkdir -p gpurun_out/r3p
cd /root/repo
timeout 1700 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r3p/pytest_gpu.log 2>&1; tail -22 gpurun_out/r3p/pytest_gpu.log
