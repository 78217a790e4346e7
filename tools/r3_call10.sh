set -x
mkdir -p gpurun_out/r3j
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q -m gpu -k "conv or fnet or pips or golden or instance_norm" > gpurun_out/r3j/pytest.log 2>&1; tail -4 gpurun_out/r3j/pytest.log
timeout 200 python tools/tracker_bench.py > gpurun_out/r3j/tracker_bench.log 2>&1; tail -2 gpurun_out/r3j/tracker_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r3j/prof_trk -o trk -- python /root/repo/tools/tracker_bench.py > /root/repo/gpurun_out/r3j/prof_trk.log 2>&1
cd /root/repo
timeout 200 python tools/forward_timeline.py > gpurun_out/r3j/timeline.log 2>&1; tail -1 gpurun_out/r3j/timeline.log | cut -c1-300
