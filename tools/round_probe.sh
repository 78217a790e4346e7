#!/bin/bash
# One gpurun call that refreshes the measurements the docs quote:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round_probe.sh r4_final'
# Results land in gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/.
set -u
TAG=${1:-probe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests -q -m gpu -s > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
timeout 120 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1
timeout 600 python bench.py > "$OUT/bench_vith.log" 2>&1
Q="--no-cpu-baseline --no-secondary"
timeout 200 python bench.py $Q --precision f16x3 --no-roofline > "$OUT/bench_vith_f16x3.log" 2>&1
timeout 200 python tools/gemm_bench.py 8 > "$OUT/gemm_microbench.log" 2>&1
timeout 200 python tools/gemm_bench.py 8 x3 > "$OUT/gemm_microbench_x3.log" 2>&1
timeout 100 python tools/attn_bench.py > "$OUT/attn_microbench.log" 2>&1; timeout 100 python tools/attn_bench.py x3 >> "$OUT/attn_microbench.log" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d "$OUT/prof" -o vith -- python "$R/bench.py" $Q --no-roofline --steps 5 --warmup 2 > "$OUT/rocprof.log" 2>&1
cd "$R"
DB=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 288 > "$OUT/vith_kernel_stats.txt" 2>&1   # 2 + 5 + 5 clips of 24 frames
rm -rf "$OUT/prof"      # the raw trace is large; the summary is what gets committed
tail -3 "$OUT/pytest_gpu.log"; tail -1 "$OUT/bench_vith.log" | cut -c1-300
