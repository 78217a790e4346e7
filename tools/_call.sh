# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r5_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_c7; mkdir -p $OUT; cd $R
# thin f32 GEMM of the tracker mixers: taller tiles / fewer workgroups (designed for 256 CUs, runs on the 32 the encoder leaves)
Q="--no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 12 --warmup 3"
run() { echo "== $1 | $2" >> $OUT/thin.log; ( export $1; timeout 120 python tools/tracker_bench.py 2>&1 | tail -1 >> $OUT/thin.log; timeout 150 python tools/forward_timeline.py 2>&1 | tail -1 | cut -c1-260 >> $OUT/thin.log; timeout 120 python bench.py $Q $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocking', d['value'], 'parity', d.get('parity',{}).get('pass'), d.get('parity',{}).get('traj_max_abs_px'))" ) >> $OUT/thin.log 2>&1; }
run "SAMPT_THIN_MIN_WGS=256" ""
run "SAMPT_THIN_MIN_WGS=128" ""
run "SAMPT_THIN_MIN_WGS=64" ""
run "SAMPT_THIN_MIN_WGS=32" ""
run "SAMPT_THIN_MIN_WGS=64 SAMPT_ENC_WGS=30" ""
run "SAMPT_THIN_MIN_WGS=32 SAMPT_ENC_WGS=30" ""
run "SAMPT_THIN_MIN_WGS=256" ""
( export SAMPT_THIN_MIN_WGS=64; timeout 300 python -m pytest tests/test_gpu_modules.py -q -k "tracker or update_window or golden" > $OUT/pytest_tracker_thin64.log 2>&1 )
cat $OUT/thin.log; tail -3 $OUT/pytest_tracker_thin64.log
