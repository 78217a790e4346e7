# scratch script of the current gpurun call: BASELINE config #5 on the final tree with the conditioned CoTracker flow head
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_final; mkdir -p $OUT; cd $R
run() { name=$1; shift; timeout 1500 python bench.py --no-secondary --no-roofline --steps 8 --warmup 3 "$@" > $OUT/bench_$name.log 2> $OUT/bench_$name.err
  tail -1 $OUT/bench_$name.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); p=d.get('parity') or {}
    print('$name', d['value'], d.get('value_pipelined'), 'parity', p.get('pass'), p.get('mask_iou_min'), p.get('masks_compared'), p.get('traj_index_identical'), p.get('traj_max_abs_px'), p.get('traj_index_differing'))
except Exception as e: print('$name failed', e)"; }
mv $OUT/bench_cfg5_hq_T64.log $OUT/bench_cfg5_hq_T64_default_head.log 2>/dev/null
run cfg5_hq_T64_conditioned --hq --tracker cotracker --square 1024 --points 16 --objects 5 --frames 64 --cotracker-delta-scale 0.001
