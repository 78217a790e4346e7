set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_final3; mkdir -p $OUT; cd $R
timeout 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_vith.log 2>&1
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python $R/tools/decode_chain_trace.py 1 > $OUT/pmc_$c.log 2>&1
  python $R/tools/pmc_summary.py $OUT/pmc_$c > $OUT/decode_chain_pmc_$c.txt 2>&1
  rm -rf $OUT/pmc_$c
done
cd $R
tail -4 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; tail -1 $OUT/bench_vith.log | cut -c80-200
tail -1 $OUT/bench_vith.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('value','ms_per_step','parity'): print(k, json.dumps(d.get(k))[:300])
print(json.dumps(d['roofline'].get('secondary'))[:1800])
"
grep -A3 "t2i_part\|fewkeys_s\|conv_f16x3" $OUT/decode_chain_pmc_FETCH_SIZE.txt | head -20
