set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2_v17; mkdir -p $OUT; cd $R
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_torchrun1.log 2>&1
echo "torchrun: $(tail -1 $OUT/bench_torchrun1.log | cut -c1-260)"
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 1 --warmup 0 --shard lpt --sequences 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_torchrun_lpt.log 2>&1
echo "lpt: $(tail -1 $OUT/bench_torchrun_lpt.log | cut -c1-400)"
