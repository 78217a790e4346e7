set -x
mkdir -p gpurun_out/r3g
cd /root/repo
for w in "28" "28,28,32" "28,30,32" "26,28,32" "29,29,32"; do
SAMPT_ENC_WGS=$w timeout 200 python tools/forward_timeline.py > gpurun_out/r3g/timeline_wgs$w.log 2>&1; echo "$w: $(tail -1 gpurun_out/r3g/timeline_wgs$w.log | cut -c1-300)"
done
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3g/bench_default.log 2>&1; tail -1 gpurun_out/r3g/bench_default.log | cut -c1-400
