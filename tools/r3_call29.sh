set -x
mkdir -p gpurun_out/r3y
cd /root/repo
timeout 900 python bench.py > gpurun_out/r3y/bench_vith.log 2>&1; tail -1 gpurun_out/r3y/bench_vith.log | cut -c1-300; grep -o '"parity": {[^}]*}' gpurun_out/r3y/bench_vith.log | cut -c1-600; grep -o '"secondary": {"unit[^}]*}' gpurun_out/r3y/bench_vith.log | cut -c1-700
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r3y/prof -o vith -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > /root/repo/gpurun_out/r3y/prof.log 2>&1
cd /root/repo; python tools/rocprof_summary.py gpurun_out/r3y/prof/vith_results.db 192 > gpurun_out/r3y/kernel_stats.txt 2>&1; head -8 gpurun_out/r3y/kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r3y/prof
