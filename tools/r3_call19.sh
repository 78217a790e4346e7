set -x
mkdir -p gpurun_out/r3s
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "flash or gemm_f16" > gpurun_out/r3s/pytest_k.log 2>&1; tail -3 gpurun_out/r3s/pytest_k.log
timeout 400 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "vit or sampt or hq" > gpurun_out/r3s/pytest_vit.log 2>&1; tail -3 gpurun_out/r3s/pytest_vit.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
timeout 200 python bench.py $B > gpurun_out/r3s/bench_dma.log 2>&1; echo "dma: $(tail -1 gpurun_out/r3s/bench_dma.log | cut -c88-140)"
SAMPT_ATTN_DMA=0 timeout 200 python bench.py $B > gpurun_out/r3s/bench_nodma.log 2>&1; echo "nodma: $(tail -1 gpurun_out/r3s/bench_nodma.log | cut -c88-140)"
timeout 200 python tools/gemm_bench.py 8 nocheck > gpurun_out/r3s/gemm.log 2>&1; tail -5 gpurun_out/r3s/gemm.log
