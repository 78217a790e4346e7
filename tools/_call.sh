set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v12; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > $OUT/pytest_attn.log 2>&1; echo "rc=$?" >> $OUT/pytest_attn.log
timeout 300 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "predict or decode or prompt or sam" > $OUT/pytest_dec.log 2>&1; echo "rc=$?" >> $OUT/pytest_dec.log
cd /tmp; export TMPDIR=/tmp
for cfg in "3 24 8 0 576 1024" "2 32 80 1 1024 1024"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace -d $OUT/prof -o dec -- python $R/tools/decode_chain_trace.py $cfg > $OUT/rocprof_$tag.log 2>&1
  DB=$(ls $OUT/prof/*/dec_results.db $OUT/prof/dec_results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py "$DB" $(echo $cfg | cut -d' ' -f1) > $OUT/decode_chain_$tag.txt 2>&1
  rm -rf $OUT/prof
done
cd $R
tail -2 $OUT/pytest_attn.log; tail -2 $OUT/pytest_dec.log; head -12 $OUT/decode_chain_3_24_8_0_576_1024.txt | cut -c1-120; head -12 $OUT/decode_chain_2_32_80_1_1024_1024.txt | cut -c1-120
