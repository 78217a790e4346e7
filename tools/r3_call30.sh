set -x
mkdir -p gpurun_out/r3z
cd /root/repo
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3z/smoke.log 2>&1; tail -2 gpurun_out/r3z/smoke.log
B="--no-cpu-baseline --no-secondary --no-roofline"
run() { n=$1; shift; timeout 240 python bench.py $B "$@" > gpurun_out/r3z/bench_$n.log 2>&1; echo "$n: $(tail -1 gpurun_out/r3z/bench_$n.log | grep -o '"value": [0-9.]*')"; }
run cfg2_vitb --model vit_b
run cfg3_cotracker --tracker cotracker --points 8 --neg-points 8 --frames 50 --steps 10 --warmup 3
run cfg4_3obj --objects 3
run cfg5_hq --hq --tracker cotracker --square 1024 --points 16 --objects 5 --steps 10 --warmup 3
run hq_pips --hq
run vith_sequential --submit sequential
