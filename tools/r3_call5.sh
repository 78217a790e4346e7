set -x
mkdir -p gpurun_out/r3e
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_f32" > gpurun_out/r3e/pytest_gemm.log 2>&1; tail -2 gpurun_out/r3e/pytest_gemm.log
timeout 600 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "pips or track or golden" > gpurun_out/r3e/pytest_pips.log 2>&1; tail -3 gpurun_out/r3e/pytest_pips.log
timeout 200 python tools/tracker_bench.py > gpurun_out/r3e/tracker_bench.log 2>&1; tail -1 gpurun_out/r3e/tracker_bench.log
SAMPT_GEMM_THIN=0 timeout 200 python tools/tracker_bench.py > gpurun_out/r3e/tracker_bench_nothin.log 2>&1; tail -1 gpurun_out/r3e/tracker_bench_nothin.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r3e/prof_trk -o trk -- python /root/repo/tools/tracker_bench.py > /root/repo/gpurun_out/r3e/prof_trk.log 2>&1
cd /root/repo
ls gpurun_out/r3e/prof_trk | head; find gpurun_out/r3e/prof_trk -name "*kernel_stats*" | head -2
f=$(find gpurun_out/r3e/prof_trk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
for w in 32 30 28; do
SAMPT_GEMM_WGS=$w timeout 200 python tools/forward_timeline.py > gpurun_out/r3e/timeline_wgs$w.log 2>&1; tail -1 gpurun_out/r3e/timeline_wgs$w.log | cut -c1-300
done
