set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v20; mkdir -p $OUT; cd $R
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-secondary --no-roofline"
timeout 100 python bench.py $Q > $OUT/bench_serial.log 2>&1; echo "serial: $(tail -1 $OUT/bench_serial.log | cut -c80-150)"
timeout 100 python bench.py $Q --dec-split 2 > $OUT/bench_split2.log 2>&1; echo "split2: $(tail -1 $OUT/bench_split2.log | cut -c80-150)"
timeout 100 python bench.py $Q --dec-pipeline > $OUT/bench_pipeline.log 2>&1; echo "pipeline: $(tail -1 $OUT/bench_pipeline.log | cut -c80-150)"
timeout 100 python bench.py $Q > $OUT/bench_serial2.log 2>&1; echo "serial: $(tail -1 $OUT/bench_serial2.log | cut -c80-150)"
