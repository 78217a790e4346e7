set -x
mkdir -p gpurun_out/r3l
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q -m gpu -k "fnet or pips or golden or resize or attention or vit" > gpurun_out/r3l/pytest.log 2>&1; tail -3 gpurun_out/r3l/pytest.log
timeout 200 python tools/tracker_bench.py > gpurun_out/r3l/tracker_bench.log 2>&1; tail -2 gpurun_out/r3l/tracker_bench.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
timeout 200 python bench.py $B > gpurun_out/r3l/bench_base.log 2>&1; echo "base: $(tail -1 gpurun_out/r3l/bench_base.log | cut -c88-140)"
timeout 200 python bench.py $B --overlap-fnet --dec-split 2 > gpurun_out/r3l/bench_ov_s2.log 2>&1; echo "ovfnet+split2: $(tail -1 gpurun_out/r3l/bench_ov_s2.log | cut -c88-140)"
SAMPT_ENC_WGS=28,28,30 timeout 200 python bench.py $B --overlap-fnet --dec-split 2 > gpurun_out/r3l/bench_ov_s2_30.log 2>&1; echo "ovfnet+split2+30: $(tail -1 gpurun_out/r3l/bench_ov_s2_30.log | cut -c88-140)"
SAMPT_ENC_WGS=28,28,30 timeout 200 python bench.py $B --overlap-fnet > gpurun_out/r3l/bench_ov_30.log 2>&1; echo "ovfnet+30: $(tail -1 gpurun_out/r3l/bench_ov_30.log | cut -c88-140)"
timeout 200 python bench.py $B --overlap-fnet > gpurun_out/r3l/bench_ov.log 2>&1; echo "ovfnet: $(tail -1 gpurun_out/r3l/bench_ov.log | cut -c88-140)"
timeout 200 python bench.py $B --dec-split 2 > gpurun_out/r3l/bench_s2.log 2>&1; echo "split2: $(tail -1 gpurun_out/r3l/bench_s2.log | cut -c88-140)"
