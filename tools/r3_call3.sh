set -x
mkdir -p gpurun_out/r3c
cd /root/repo
for w in 32 30 28; do
SAMPT_GEMM_WGS=$w timeout 200 python tools/forward_timeline.py > gpurun_out/r3c/timeline_wgs$w.log 2>&1; tail -2 gpurun_out/r3c/timeline_wgs$w.log | cut -c1-300
done
SAMPT_GEMM_P8=0 timeout 200 python tools/forward_timeline.py > gpurun_out/r3c/timeline_old.log 2>&1; tail -2 gpurun_out/r3c/timeline_old.log | cut -c1-300
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
for w in 30 28; do
SAMPT_GEMM_WGS=$w timeout 200 python bench.py $B > gpurun_out/r3c/bench_wgs$w.log 2>&1; tail -1 gpurun_out/r3c/bench_wgs$w.log | cut -c80-200
done
timeout 1500 python -m pytest tests/test_gpu_bench_parity.py -x -q -m gpu -s > gpurun_out/r3c/pytest_parity.log 2>&1; tail -5 gpurun_out/r3c/pytest_parity.log; grep "parity\]" gpurun_out/r3c/pytest_parity.log | cut -c1-400
