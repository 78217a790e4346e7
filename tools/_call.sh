# scratch script of the current gpurun call: flash attention with more queries per workgroup (global: 6 / 8 waves, windowed: 7) —
# kernel tests under every setting, micro-benchmark, bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c10; mkdir -p $OUT; cd $R
for w in "4,4" "6,4" "8,4" "4,7"; do
  SAMPT_ATTN_WAVES=$w timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "flash or window_attention or vit_attention" > $OUT/pytest_attn_$w.log 2>&1; echo "waves=$w: $(tail -1 $OUT/pytest_attn_$w.log)"
  SAMPT_ATTN_WAVES=$w timeout 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/attn_bench_$w.log; cat $OUT/attn_bench_$w.log
done
for w in "4,4" "6,4" "8,4" "6,7"; do
  SAMPT_ATTN_WAVES=$w timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 10 --warmup 3 > $OUT/bench_$w.log 2>&1
  tail -1 $OUT/bench_$w.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves=$w', d['value'], d.get('timeline'), d['parity']['mask_iou_min'], d['parity']['pass'])"
done
