set -x
mkdir -p gpurun_out/r3d
cd /root/repo
ls -la --time-style=full-iso sam_pt_amd/libsampt_hip.so
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > gpurun_out/r3d/pytest_gemm.log 2>&1; tail -3 gpurun_out/r3d/pytest_gemm.log
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "pips or track or sampt or prefetch or fnet or update or reinit or golden" > gpurun_out/r3d/pytest_pips.log 2>&1; tail -5 gpurun_out/r3d/pytest_pips.log
timeout 200 python tools/stage_times.py > gpurun_out/r3d/stage_times.log 2>&1; tail -1 gpurun_out/r3d/stage_times.log
for w in 32 30 29 28 26; do
SAMPT_GEMM_WGS=$w timeout 200 python tools/forward_timeline.py > gpurun_out/r3d/timeline_wgs$w.log 2>&1; tail -1 gpurun_out/r3d/timeline_wgs$w.log | cut -c1-300
done
