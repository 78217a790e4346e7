# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c4; mkdir -p $OUT; cd $R
L=sam_pt_amd/libsampt_hip.so; cp $L /tmp/ilv.so
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', d['value'], d['value_per_forward'], 'insitu', r['achieved'], 'iso', r['isolated_achieved'])"; }
for rep in 1 2; do
  cp /tmp/ilv.so $L; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_ilv_$rep.log 2>&1; show $OUT/bench_ilv_$rep.log ilv
  cp sam_pt_amd/libsampt_hip_noilv.so $L; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_noilv_$rep.log 2>&1; show $OUT/bench_noilv_$rep.log noilv
done
cp sam_pt_amd/libsampt_hip_noilv.so $L; timeout 200 python tools/gemm_bench.py 8 nocheck > $OUT/gemm_noilv.log 2>&1; tail -11 $OUT/gemm_noilv.log
cp /tmp/ilv.so $L; timeout 200 python tools/gemm_bench.py 8 nocheck > $OUT/gemm_ilv.log 2>&1; tail -11 $OUT/gemm_ilv.log
