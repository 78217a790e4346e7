set -x
mkdir -p gpurun_out/r3f
cd /root/repo
timeout 100 python tools/thin_bench.py > gpurun_out/r3f/thin_default.log 2>&1; tail -3 gpurun_out/r3f/thin_default.log
for cfg in "1 4" "1 8" "1 16" "2 4" "2 8" "2 16" "4 8" "4 16"; do set -- $cfg
SAMPT_THIN_FM=$1 SAMPT_THIN_NWV=$2 timeout 100 python tools/thin_bench.py 8 > gpurun_out/r3f/thin_fm$1_nwv$2.log 2>&1; echo "FM=$1 NWV=$2: $(tail -1 gpurun_out/r3f/thin_fm$1_nwv$2.log)"
done
timeout 200 python tools/tracker_bench.py > gpurun_out/r3f/tracker_bench.log 2>&1; tail -1 gpurun_out/r3f/tracker_bench.log
timeout 200 python tools/tracker_bench.py --objects 3 > gpurun_out/r3f/tracker_bench_3obj.log 2>&1; tail -1 gpurun_out/r3f/tracker_bench_3obj.log
for w in 32 30 28; do
SAMPT_GEMM_WGS=$w timeout 200 python tools/forward_timeline.py > gpurun_out/r3f/timeline_wgs$w.log 2>&1; tail -1 gpurun_out/r3f/timeline_wgs$w.log | cut -c1-300
done
