set -x
mkdir -p gpurun_out/r3i
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_modules.py tests/test_gpu_kernels.py -x -q -m gpu -k "fnet or conv or pips or track or golden or instance_norm or sampt" > gpurun_out/r3i/pytest.log 2>&1; tail -3 gpurun_out/r3i/pytest.log
timeout 200 python tools/tracker_bench.py > gpurun_out/r3i/tracker_bench.log 2>&1; tail -2 gpurun_out/r3i/tracker_bench.log
SAMPT_FNET_PLANES=0 timeout 200 python tools/tracker_bench.py > gpurun_out/r3i/tracker_bench_noplanes.log 2>&1; tail -2 gpurun_out/r3i/tracker_bench_noplanes.log
timeout 200 python tools/forward_timeline.py > gpurun_out/r3i/timeline.log 2>&1; tail -1 gpurun_out/r3i/timeline.log | cut -c1-300
