# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r5_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_c6; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_modules.py -q -s -k "frame_sharded_emulation" > $OUT/pytest_sharded_emulation.log 2>&1
timeout 200 python tools/forward_timeline.py > $OUT/timeline.log 2>&1
timeout 200 python tools/stage_times.py > $OUT/stage_times.log 2>&1
timeout 400 python bench.py --emulate-ranks 2,4,8 --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 6 --warmup 2 > $OUT/bench_emulate.log 2>&1
timeout 200 python tools/tracker_bench.py > $OUT/tracker_bench.log 2>&1
tail -3 $OUT/pytest_sharded_emulation.log; tail -3 $OUT/timeline.log; tail -1 $OUT/stage_times.log; tail -1 $OUT/bench_emulate.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); m=d['frame_sharding_model']; print(m['one_gpu_ms_per_clip'], {k:(v['predicted_ms_per_clip'], v['predicted_speedup']) for k,v in m['by_world'].items()})"; tail -4 $OUT/tracker_bench.log
