#!/bin/bash
# One gpurun call that refreshes the measurements the docs quote:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/round_probe.sh r2_v0'
# Results land in gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/.
set -u
TAG=${1:-probe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 420 python -m pytest tests -q -m gpu -s > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
timeout 300 python bench.py > "$OUT/bench_vith.log" 2>&1
Q="--no-cpu-baseline --no-roofline --no-secondary"
SAMPT_DEC_F16X3=1 timeout 90 python bench.py $Q > "$OUT/bench_vith_dec_f16x3.log" 2>&1
# (round-2 experiment SAMPT_GEMM_LDS_PAD removed with the experimental GEMM variants)
SAMPT_PIPS_FUSE_REDUCE=1 timeout 90 python bench.py $Q > "$OUT/bench_vith_fuse_reduce.log" 2>&1
SAMPT_TEST_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_gpu_modules.py -q -m gpu -s -k "opt_in" > "$OUT/pytest_experimental.log" 2>&1
timeout 90 python tools/stage_times.py > "$OUT/stage_times.log" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace -d "$OUT/prof" -o vith -- python "$R/bench.py" $Q > "$OUT/rocprof.log" 2>&1
cd "$R"
DB=$(ls "$OUT"/prof/*/vith_results.db "$OUT"/prof/vith_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 168 > "$OUT/vith_kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"      # the raw trace is large; the summary is what gets committed
tail -3 "$OUT/pytest_gpu.log"; tail -1 "$OUT/bench_vith.log" | cut -c1-300
