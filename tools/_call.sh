# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c27; mkdir -p $OUT; cd $R
Q="--no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 5"
run() { echo "== SAMPT_ENC_WGS=${1:-default}" >> $OUT/enc_wgs.log; ( [ -n "${1:-}" ] && export SAMPT_ENC_WGS=$1; timeout 45 python bench.py $Q 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_per_forward'], d.get('parity',{}).get('pass'))" ) >> $OUT/enc_wgs.log 2>&1; }
run ""; run 28,28,32; run ""
cat $OUT/enc_wgs.log
