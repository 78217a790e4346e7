# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c2; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm or layernorm or flash or kmedoids" > $OUT/pytest_kernels.log 2>&1; tail -4 $OUT/pytest_kernels.log
timeout 500 python -m pytest tests/test_gpu_modules.py -q -m gpu -s -k "vit_test_encoder or dead_row or prefetch or frame_sharded" > $OUT/pytest_vit.log 2>&1; tail -4 $OUT/pytest_vit.log
timeout 200 python tools/p8_diag.py > $OUT/p8_diag.log 2>&1; cat $OUT/p8_diag.log
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -q -m gpu -s -k "bench_clip" > $OUT/pytest_parity.log 2>&1; grep "bench parity\|passed\|failed" $OUT/pytest_parity.log | cut -c1-700
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --emulate-ranks 2,4,8 > $OUT/bench_emulate.log 2>&1; tail -1 $OUT/bench_emulate.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('frame_sharding_model'))[:3000])"
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --emulate-ranks 8 --pips-vis-bias 4.0 > $OUT/bench_emulate_vb4.log 2>&1; tail -1 $OUT/bench_emulate_vb4.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d.get('frame_sharding_model'))[:1500])"
