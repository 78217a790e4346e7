# scratch script of the current gpurun call: token -> image attention with 16 queries per workgroup (K / V streamed once per frame)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c39; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "attention" > $OUT/pytest_attn.log 2>&1; tail -3 $OUT/pytest_attn.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_bench_parity.py -x -q -k "dec or sam or golden or predictor or parity or stream" > $OUT/pytest_mod.log 2>&1; tail -3 $OUT/pytest_mod.log | cut -c1-300
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $OUT/prof -o clip -- python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 4 --warmup 2 > $OUT/rocprof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python $R/tools/rocprof_by_grid.py "$DB" "" 6 > $OUT/clip_by_grid.txt 2>&1; rm -rf $OUT/prof
grep "t2i\|fewkeys" $OUT/clip_by_grid.txt | cut -c1-140
tail -1 $OUT/rocprof.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(d['value'], d.get('timeline'))"
