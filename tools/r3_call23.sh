set -x
mkdir -p gpurun_out/r3w
cd /root/repo
timeout 400 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "stream_of_clips or pipelined_decoder or sampt_end_to_end or ragged" > gpurun_out/r3w/pytest_stream.log 2>&1; tail -5 gpurun_out/r3w/pytest_stream.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline"
timeout 200 python bench.py $B > gpurun_out/r3w/bench_pipelined.log 2>&1; echo "pipelined: $(tail -1 gpurun_out/r3w/bench_pipelined.log | cut -c88-140)"
timeout 200 python bench.py $B --submit sequential > gpurun_out/r3w/bench_sequential.log 2>&1; echo "sequential: $(tail -1 gpurun_out/r3w/bench_sequential.log | cut -c88-140)"
timeout 200 python bench.py $B > gpurun_out/r3w/bench_pipelined2.log 2>&1; echo "pipelined: $(tail -1 gpurun_out/r3w/bench_pipelined2.log | cut -c88-140)"
SAMPT_ENC_WGS=30 timeout 200 python bench.py $B > gpurun_out/r3w/bench_pipelined_wgs30.log 2>&1; echo "pipelined wgs30: $(tail -1 gpurun_out/r3w/bench_pipelined_wgs30.log | cut -c88-140)"
