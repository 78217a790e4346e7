# scratch script of the current gpurun call: why does a 12-frame encoder batch starve the window chain?  kernel traces grouped by grid
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c12; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for b in 12 8; do
  SAMPT_ENC_WGS=30 timeout 400 rocprofv3 --kernel-trace -d $OUT/prof_b$b -o clip -- python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 3 --warmup 2 --encode-batch $b > $OUT/rocprof_b$b.log 2>&1
  DB=$(find $OUT/prof_b$b -name "*.db" | head -1)
  python $R/tools/rocprof_by_grid.py "$DB" "" 8 > $OUT/clip_by_grid_b$b.txt 2>&1
  python $R/tools/rocprof_sequence.py "$DB" 1500 > $OUT/clip_sequence_b$b.txt 2>&1
  rm -rf $OUT/prof_b$b
  echo "== batch $b"; grep -E "pips_mix|gemm_f16_p8|k_flash|layernorm_rows_v4<5>" $OUT/clip_by_grid_b$b.txt | head -12
done
