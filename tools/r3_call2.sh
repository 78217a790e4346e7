set -x
mkdir -p gpurun_out/r3b
cd /root/repo
timeout 150 python tools/gemm_bench.py 8 > gpurun_out/r3b/p8_b8.log 2>&1; echo rc=$?
timeout 100 python tools/gemm_bench.py 24 nocheck > gpurun_out/r3b/p8_b24.log 2>&1; echo rc=$?
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_f16" > gpurun_out/r3b/pytest_gemm.log 2>&1; tail -3 gpurun_out/r3b/pytest_gemm.log
timeout 400 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "vit" > gpurun_out/r3b/pytest_vit.log 2>&1; tail -3 gpurun_out/r3b/pytest_vit.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
timeout 200 python bench.py $B > gpurun_out/r3b/bench_p8_b8.log 2>&1; tail -1 gpurun_out/r3b/bench_p8_b8.log | cut -c1-250
timeout 200 python bench.py $B --encode-batch 24 > gpurun_out/r3b/bench_p8_b24.log 2>&1; tail -1 gpurun_out/r3b/bench_p8_b24.log | cut -c1-250
timeout 200 python bench.py $B --encode-batch 12 > gpurun_out/r3b/bench_p8_b12.log 2>&1; tail -1 gpurun_out/r3b/bench_p8_b12.log | cut -c1-250
SAMPT_GEMM_P8=0 timeout 200 python bench.py $B > gpurun_out/r3b/bench_old_b8.log 2>&1; tail -1 gpurun_out/r3b/bench_old_b8.log | cut -c1-250
for f in p8_b8 p8_b24; do echo "== $f"; grep -v "^check\|amdgpu.ids" gpurun_out/r3b/$f.log | tail -12; done
