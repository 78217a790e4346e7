set -x
R=/root/repo
mkdir -p $R/gpurun_out/r3pmc
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3pmc/sq1 -- python $R/tools/attn_bench.py > $R/gpurun_out/r3pmc/pmc_sq1.log 2>&1; tail -2 $R/gpurun_out/r3pmc/pmc_sq1.log
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/r3pmc/sq2 -- python $R/tools/attn_bench.py > $R/gpurun_out/r3pmc/pmc_sq2.log 2>&1; tail -2 $R/gpurun_out/r3pmc/pmc_sq2.log
cd $R
(python tools/pmc_summary.py gpurun_out/r3pmc/sq1 flash; python tools/pmc_summary.py gpurun_out/r3pmc/sq2 flash) > gpurun_out/r3pmc/attn_sq_counters.txt 2>&1; cat gpurun_out/r3pmc/attn_sq_counters.txt | cut -c1-160
rm -rf gpurun_out/r3pmc/sq1 gpurun_out/r3pmc/sq2
