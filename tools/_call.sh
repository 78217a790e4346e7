# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_c11; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"
timeout 250 rocprofv3 --pmc $C1 --output-format csv -d $OUT/g1 -- python $R/tools/gemm_bench.py 8 nocheck > $OUT/pmc_g1.log 2>&1
timeout 250 rocprofv3 --pmc $C2 --output-format csv -d $OUT/g2 -- python $R/tools/gemm_bench.py 8 nocheck > $OUT/pmc_g2.log 2>&1
timeout 300 rocprofv3 --pmc $C1 --output-format csv -d $OUT/g3 -- python $R/tools/gemm_bench.py 8 x3 > $OUT/pmc_g3.log 2>&1
cd $R
(echo "# rocprofv3 --pmc passes over tools/gemm_bench.py 8 (fp16: g1 / g2) and tools/gemm_bench.py 8 x3 (g3): per-kernel means over all launches of the run"; python tools/pmc_summary.py $OUT/g1 gemm_f16; python tools/pmc_summary.py $OUT/g2 gemm_f16; python tools/pmc_summary.py $OUT/g3 gemm_f16) > $OUT/gemm_sq_counters.txt 2>&1; cut -c1-110 $OUT/gemm_sq_counters.txt
rm -rf $OUT/g1 $OUT/g2 $OUT/g3
