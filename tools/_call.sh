set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c45; mkdir -p $OUT; cd $R
timeout 1500 python bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-roofline --no-pipelined --emulate-ranks 2,4,8 > $OUT/bench_emulate.log 2> $OUT/bench_emulate.err
tail -1 $OUT/bench_emulate.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['frame_sharding_model']; print(d['value']); print({k:(v['predicted_ms_per_clip'], v['predicted_speedup']) for k,v in m['by_world'].items()}); print({k:v for k,v in m.items() if k!='by_world'})"
