set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v3; mkdir -p $OUT; cd $R
timeout 500 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_modules.py -q -m gpu -s -k "bench_clip or graph or pipelined or predict_torch or hq_sam or vit_b_encoder or vit_test_encoder" > $OUT/pytest_subset.log 2>&1
Q="--no-cpu-baseline --no-secondary"
timeout 150 python bench.py $Q > $OUT/bench_default.log 2>&1
timeout 120 python bench.py $Q --no-roofline --no-dec-graph > $OUT/bench_nograph.log 2>&1
timeout 120 python bench.py $Q --no-roofline --dec-split 2 > $OUT/bench_split2.log 2>&1
SAMPT_DEC_F16X3=0 timeout 120 python bench.py $Q --no-roofline > $OUT/bench_dec_f32.log 2>&1
timeout 120 python tools/graph_vs_eager.py > $OUT/graph_vs_eager.log 2>&1
timeout 90 python tools/stage_times.py > $OUT/stage_times.log 2>&1
tail -3 $OUT/pytest_subset.log; grep "bench parity" $OUT/pytest_subset.log | cut -c1-200; for f in default nograph split2 dec_f32; do echo $f; tail -1 $OUT/bench_$f.log | cut -c1-200; done; cat $OUT/graph_vs_eager.log | tail -6; tail -1 $OUT/stage_times.log
