set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v4; mkdir -p $OUT; cd $R
timeout 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm_f16" > $OUT/pytest_gemm_default.log 2>&1
SAMPT_GEMM_BN160=1 timeout 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm_f16" > $OUT/pytest_gemm_bn160.log 2>&1
timeout 120 python tools/gemm_bench.py 8 > $OUT/gemm_microbench_default.log 2>&1
SAMPT_GEMM_BN160=1 timeout 120 python tools/gemm_bench.py 8 > $OUT/gemm_microbench_bn160.log 2>&1
timeout 500 python -m pytest tests/test_gpu_modules.py tests/test_gpu_cotracker.py -q -m gpu -s -k "tracker or end_to_end or pipelined or golden or short_clip" > $OUT/pytest_subset.log 2>&1
Q="--no-cpu-baseline --no-secondary"
timeout 200 python bench.py $Q > $OUT/bench_default.log 2>&1
SAMPT_GEMM_BN160=1 timeout 200 python bench.py $Q > $OUT/bench_bn160.log 2>&1
timeout 120 python bench.py $Q --no-roofline --pips-vis-bias 4.0 > $OUT/bench_visbias4.log 2>&1
timeout 120 python bench.py $Q --no-roofline --objects 3 > $OUT/bench_cfg4_3obj.log 2>&1
timeout 90 python tools/stage_times.py > $OUT/stage_times.log 2>&1
tail -2 $OUT/pytest_gemm_default.log; tail -2 $OUT/pytest_gemm_bn160.log; paste $OUT/gemm_microbench_default.log $OUT/gemm_microbench_bn160.log | cut -c1-200
tail -3 $OUT/pytest_subset.log; for f in default bn160 visbias4 cfg4_3obj; do echo $f; tail -1 $OUT/bench_$f.log | cut -c1-200; done; tail -1 $OUT/stage_times.log
tail -1 $OUT/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d['roofline'][k] for k in ('achieved','frac','avg_launch_us')})
for e in d['roofline'].get('secondary',[]): print(e)
" 2>&1 | tail -8
