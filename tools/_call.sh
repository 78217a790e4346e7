set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_final2; mkdir -p $OUT; cd $R
timeout 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_vith.log 2>&1
Q="--no-cpu-baseline --no-secondary --no-roofline"
timeout 200 python bench.py $Q --hq --tracker cotracker --points 16 --objects 5 --square 1024 --frames 24 > $OUT/bench_cfg5_hq_cotracker.log 2>&1
timeout 200 python bench.py $Q --tracker cotracker --points 8 --neg-points 8 --frames 50 > $OUT/bench_cfg3_cotracker.log 2>&1
timeout 200 python bench.py $Q --objects 3 > $OUT/bench_cfg4_3obj.log 2>&1
timeout 200 python bench.py $Q --hq > $OUT/bench_hq_pips.log 2>&1
timeout 200 python bench.py $Q --model vit_b > $OUT/bench_vitb.log 2>&1
tail -4 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; for f in bench_vith bench_cfg5_hq_cotracker bench_cfg3_cotracker bench_cfg4_3obj bench_hq_pips bench_vitb; do echo $f; tail -1 $OUT/$f.log | cut -c80-200; done
tail -1 $OUT/bench_vith.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('value','ms_per_step','roofline','secondary','parity','cpu_baseline'): print(k, json.dumps(d.get(k))[:500])
"
