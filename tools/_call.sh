# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r5_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_c1; mkdir -p $OUT; cd $R
# 1. new / changed tests of this round
timeout 500 python -m pytest tests/test_gpu_kernels.py -q -s -k "erode_device or shi_tomasi or vos_index or kmedoids" > $OUT/pytest_new_kernels.log 2>&1
timeout 400 python -m pytest tests/test_gpu_modules.py -q -s -k "bias_correction or vit_b_encoder or dead_row" > $OUT/pytest_bias.log 2>&1
timeout 500 python -m pytest tests/test_gpu_dist_nccl.py tests/test_gpu_cotracker.py -q -s -k "rccl or bench_under or long_clip" > $OUT/pytest_dist_cotracker.log 2>&1
# 2. the default bench line with the new semantics (value = blocking forward over two alternating clips), cached oracle
timeout 400 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_default.log 2>&1
Q="--no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 5"
( export SAMPT_VIT_BIAS_CORR=0; timeout 120 python bench.py $Q > $OUT/bench_no_bias_corr.log 2>&1 )
# 3. persistent GEMM workgroups per launch kind (qkv/proj/fc1/fc2)
run() { echo "== SAMPT_ENC_WGS=$1" >> $OUT/enc_wgs_kind.log; ( export SAMPT_ENC_WGS=$1; timeout 120 python bench.py $Q 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocking', d['value'], 'pipelined', d['value_pipelined'], 'parity', d.get('parity',{}).get('pass'), d.get('parity',{}).get('mask_iou_min'))" ) >> $OUT/enc_wgs_kind.log 2>&1; }
run 28; run 30/28/30/28; run 30/27/30/27; run 32/28/32/28; run 29/28/29/28; run 30; run 28
# 4. vendor yardstick on the same shapes
timeout 200 python tools/blas_ceiling.py 8 > $OUT/blas_ceiling.log 2>&1
tail -3 $OUT/pytest_new_kernels.log $OUT/pytest_bias.log $OUT/pytest_dist_cotracker.log; cat $OUT/enc_wgs_kind.log; tail -1 $OUT/bench_default.log | cut -c1-600; tail -1 $OUT/bench_no_bias_corr.log | cut -c1-400
