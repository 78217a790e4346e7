set -x
mkdir -p gpurun_out/r3r
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "flash" > gpurun_out/r3r/pytest_flash.log 2>&1; tail -3 gpurun_out/r3r/pytest_flash.log
timeout 300 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "vit" > gpurun_out/r3r/pytest_vit.log 2>&1; tail -3 gpurun_out/r3r/pytest_vit.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
timeout 200 python bench.py $B > gpurun_out/r3r/bench_dma.log 2>&1; echo "dma: $(tail -1 gpurun_out/r3r/bench_dma.log | cut -c88-140)"
SAMPT_ATTN_DMA=0 timeout 200 python bench.py $B > gpurun_out/r3r/bench_nodma.log 2>&1; echo "nodma: $(tail -1 gpurun_out/r3r/bench_nodma.log | cut -c88-140)"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r3r/prof -o vith -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > /root/repo/gpurun_out/r3r/prof.log 2>&1
cd /root/repo; python tools/rocprof_summary.py gpurun_out/r3r/prof/vith_results.db 144 > gpurun_out/r3r/kernel_stats.txt 2>&1; head -30 gpurun_out/r3r/kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r3r/prof
