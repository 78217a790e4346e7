set -x
mkdir -p gpurun_out/r3k
cd /root/repo
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
timeout 200 python bench.py $B > gpurun_out/r3k/bench_base.log 2>&1; echo "base: $(tail -1 gpurun_out/r3k/bench_base.log | cut -c88-140)"
timeout 200 python bench.py $B --dec-split 1 > gpurun_out/r3k/bench_split1.log 2>&1; echo "split1: $(tail -1 gpurun_out/r3k/bench_split1.log | cut -c88-140)"
timeout 200 python bench.py $B --dec-split 2 > gpurun_out/r3k/bench_split2.log 2>&1; echo "split2: $(tail -1 gpurun_out/r3k/bench_split2.log | cut -c88-140)"
timeout 200 python bench.py $B --dec-pipeline > gpurun_out/r3k/bench_pipe.log 2>&1; echo "pipe: $(tail -1 gpurun_out/r3k/bench_pipe.log | cut -c88-140)"
SAMPT_ENC_WGS=28,28,30 timeout 200 python bench.py $B > gpurun_out/r3k/bench_282830.log 2>&1; echo "28,28,30: $(tail -1 gpurun_out/r3k/bench_282830.log | cut -c88-140)"
timeout 200 python bench.py $B --overlap-fnet > gpurun_out/r3k/bench_ovfnet.log 2>&1; echo "overlap-fnet: $(tail -1 gpurun_out/r3k/bench_ovfnet.log | cut -c88-140)"
