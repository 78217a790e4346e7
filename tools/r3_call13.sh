set -x
R=/root/repo
mkdir -p $R/gpurun_out/r3m
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/r3m/hbm_$c -- python $R/tools/gemm_bench.py 8 nocheck > $R/gpurun_out/r3m/pmc_$c.log 2>&1; tail -2 $R/gpurun_out/r3m/pmc_$c.log
done
SAMPT_GEMM_R=8 timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3m/hbm_FETCH_SIZE_R8 -- python $R/tools/gemm_bench.py 8 nocheck > $R/gpurun_out/r3m/pmc_FETCH_R8.log 2>&1
SAMPT_GEMM_R=2 timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3m/hbm_FETCH_SIZE_R2 -- python $R/tools/gemm_bench.py 8 nocheck > $R/gpurun_out/r3m/pmc_FETCH_R2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3m/sq1 -- python $R/tools/gemm_bench.py 8 nocheck > $R/gpurun_out/r3m/pmc_sq1.log 2>&1; tail -1 $R/gpurun_out/r3m/pmc_sq1.log
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/r3m/sq2 -- python $R/tools/gemm_bench.py 8 nocheck > $R/gpurun_out/r3m/pmc_sq2.log 2>&1; tail -1 $R/gpurun_out/r3m/pmc_sq2.log
cd $R
python tools/gemm_traffic.py gpurun_out/r3m/hbm_FETCH_SIZE gpurun_out/r3m/hbm_WRITE_SIZE 8 > gpurun_out/r3m/gemm_hbm_traffic.json; head -30 gpurun_out/r3m/gemm_hbm_traffic.json
python tools/gemm_traffic.py gpurun_out/r3m/hbm_FETCH_SIZE_R8 gpurun_out/r3m/hbm_WRITE_SIZE 8 > gpurun_out/r3m/gemm_hbm_traffic_R8.json
python tools/gemm_traffic.py gpurun_out/r3m/hbm_FETCH_SIZE_R2 gpurun_out/r3m/hbm_WRITE_SIZE 8 > gpurun_out/r3m/gemm_hbm_traffic_R2.json
python tools/pmc_summary.py gpurun_out/r3m/sq1 gemm_f16 > gpurun_out/r3m/sq1.txt; python tools/pmc_summary.py gpurun_out/r3m/sq2 gemm_f16 > gpurun_out/r3m/sq2.txt
cat gpurun_out/r3m/sq1.txt gpurun_out/r3m/sq2.txt | cut -c1-150
rm -rf gpurun_out/r3m/hbm_* gpurun_out/r3m/sq1 gpurun_out/r3m/sq2
