set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v2; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests/test_gpu_modules.py -q -m gpu -s -k "prompt_size or graph or pipelined or ragged or end_to_end or hq_sampt" > $OUT/pytest_modules.log 2>&1
Q="--no-cpu-baseline --no-secondary"
timeout 150 python bench.py $Q > $OUT/bench_default.log 2>&1
timeout 120 python bench.py $Q --no-roofline --no-dec-graph > $OUT/bench_nograph.log 2>&1
timeout 120 python bench.py $Q --no-roofline --no-dec-pipeline > $OUT/bench_serial.log 2>&1
timeout 120 python bench.py $Q --no-roofline --model vit_b > $OUT/bench_cfg2_vitb.log 2>&1
timeout 200 python bench.py $Q --no-roofline --tracker cotracker --neg-points 8 --frames 50 > $OUT/bench_cfg3_cotracker.log 2>&1
timeout 200 python bench.py $Q --no-roofline --hq --tracker cotracker --points 16 --objects 5 --square 1024 --frames 24 > $OUT/bench_cfg5_hq_cotracker.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace -d "$OUT/prof" -o vith -- python "$R/bench.py" $Q --no-roofline > "$OUT/rocprof.log" 2>&1
cd "$R"
DB=$(ls "$OUT"/prof/*/vith_results.db "$OUT"/prof/vith_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" 168 > "$OUT/vith_kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
tail -3 $OUT/pytest_modules.log; for f in default nograph serial cfg2_vitb cfg3_cotracker cfg5_hq_cotracker; do echo $f; tail -1 $OUT/bench_$f.log | cut -c1-230; done
