set -x
mkdir -p gpurun_out/r3t
cd /root/repo
timeout 200 python tools/forward_timeline.py > gpurun_out/r3t/timeline_dma.log 2>&1; tail -1 gpurun_out/r3t/timeline_dma.log | cut -c1-300
SAMPT_ATTN_DMA=0 timeout 200 python tools/forward_timeline.py > gpurun_out/r3t/timeline_nodma.log 2>&1; tail -1 gpurun_out/r3t/timeline_nodma.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r3t/prof -o vith -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > /root/repo/gpurun_out/r3t/prof.log 2>&1
cd /root/repo; python tools/rocprof_summary.py gpurun_out/r3t/prof/vith_results.db 144 > gpurun_out/r3t/kernel_stats.txt 2>&1; head -24 gpurun_out/r3t/kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r3t/prof
