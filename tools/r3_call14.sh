set -x
mkdir -p gpurun_out/r3n
cd /root/repo
timeout 200 python __graft_entry__.py smoke > gpurun_out/r3n/smoke.log 2>&1; tail -2 gpurun_out/r3n/smoke.log
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r3n/bench_vith.log 2>&1; tail -1 gpurun_out/r3n/bench_vith.log | cut -c1-300; grep -o '"parity": {[^}]*}' gpurun_out/r3n/bench_vith.log | cut -c1-600
