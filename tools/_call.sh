# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c25; mkdir -p $OUT; cd $R
for d in 0 1 2 3 0; do echo "== diag=$d (0 shipped, 1 prologue only, 2 tile loop without MFMA / softmax, 3 no rel-pos table MFMAs)" >> $OUT/attn_diag.log
  ATTN_BENCH_LIB=tools/_ab/libsampt_hip_diag.so SAMPT_FLASH_DIAG=$d timeout 60 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids >> $OUT/attn_diag.log; done
cat $OUT/attn_diag.log
