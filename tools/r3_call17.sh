set -x
mkdir -p gpurun_out/r3q
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "vit_b_encoder" > gpurun_out/r3q/pytest_vitb.log 2>&1; tail -2 gpurun_out/r3q/pytest_vitb.log
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r3q/bench_vith.log 2>&1; tail -1 gpurun_out/r3q/bench_vith.log | cut -c1-260; grep -o '"parity": {[^}]*}' gpurun_out/r3q/bench_vith.log | cut -c1-700
