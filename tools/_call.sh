# scratch script of the current gpurun call: sequence-sharded scaling predicted from measured per-sequence times (bench.py --lpt-model)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c44; mkdir -p $OUT; cd $R
timeout 1500 python bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-roofline --no-pipelined --lpt-model 2,4,8 > $OUT/bench_lpt_model.log 2> $OUT/bench_lpt_model.err
tail -1 $OUT/bench_lpt_model.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['sequence_sharding_model']; print(d['value']); print(json.dumps(m)[:1800])"
