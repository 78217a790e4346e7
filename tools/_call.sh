# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r5_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_c10; mkdir -p $OUT; cd $R
Q="--no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 5"
run() { echo "== THIN=$1 args: $2" >> $OUT/split.log; ( export SAMPT_THIN_MIN_WGS=$1; timeout 120 python bench.py $Q $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocking', d['value'], 'pipelined', d['value_pipelined'], 'parity', d.get('parity',{}).get('pass'))" ) >> $OUT/split.log 2>&1; }
run 256 ""
run 64 ""
run 64 "--dec-split 2"
run 64 "--dec-split 1"
run 256 "--dec-split 2"
run 256 ""
run 64 "--dec-split 2"
run 64 ""
cat $OUT/split.log
