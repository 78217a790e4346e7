set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v5; mkdir -p $OUT; cd $R
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 400 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "tracker or golden or short_clip or end_to_end or reinit or single_frame" > $OUT/pytest_tracker.log 2>&1
Q="--no-cpu-baseline --no-secondary"
timeout 200 python bench.py $Q > $OUT/bench_default.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace -d "$OUT/prof" -o vith -- python "$R/bench.py" $Q --no-roofline > "$OUT/rocprof.log" 2>&1
DB=$(ls "$OUT"/prof/*/vith_results.db "$OUT"/prof/vith_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py "$DB" 168 > "$OUT/vith_kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/hbm_FETCH -- python $R/tools/gemm_bench.py 8 > $OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/hbm_WRITE -- python $R/tools/gemm_bench.py 8 > $OUT/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python $R/tools/gemm_bench.py 8 > $OUT/pmc_sq.log 2>&1
cd $R
python tools/gemm_traffic.py $OUT/hbm_FETCH $OUT/hbm_WRITE 8 > $OUT/r2_gemm_hbm_traffic.json 2> $OUT/gemm_traffic.err
python tools/pmc_summary.py $OUT/sq gemm_f16 > $OUT/r2_gemm_sq_counters.txt 2>&1
rm -rf $OUT/hbm_FETCH $OUT/hbm_WRITE $OUT/sq
tail -2 $OUT/smoke.log; tail -2 $OUT/pytest_tracker.log; tail -1 $OUT/bench_default.log | cut -c1-200; head -12 $OUT/vith_kernel_stats.txt; head -30 $OUT/r2_gemm_hbm_traffic.json; cat $OUT/r2_gemm_sq_counters.txt | head -20; tail -3 $OUT/gemm_traffic.err
