# scratch script of the current gpurun call: full GPU suite + bench + kernel trace by grid on the current tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c34; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --steps 8 --warmup 3 --no-secondary --no-cpu-baseline --no-roofline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print(d["value"], d.get("value_pipelined"), d.get("timeline"), d.get("chain_launches_per_round"))
PY
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof -o clip -- python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 3 --warmup 2 > $OUT/rocprof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python $R/tools/rocprof_by_grid.py "$DB" "" 8 > $OUT/clip_by_grid.txt 2>&1
python $R/tools/rocprof_sequence.py "$DB" 1200 > $OUT/clip_sequence.txt 2>&1
rm -rf $OUT/prof
head -30 $OUT/clip_by_grid.txt | cut -c1-150
