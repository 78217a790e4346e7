set -x
mkdir -p gpurun_out/r3o
cd /root/repo
free -g | head -3; nproc; cat /proc/meminfo | head -3
(timeout 300 python __graft_entry__.py smoke > gpurun_out/r3o/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3o/smoke.log)
free -g | head -2
# the bench line with the CPU leg on 3 frames (the round-2 default), memory-capped, RSS sampled
( while true; do ps -eo rss,comm --sort=-rss | head -2 | tail -1; sleep 5; done ) > gpurun_out/r3o/rss.log 2>&1 &
MON=$!
(ulimit -v $((200*1024*1024)); timeout 900 python bench.py --steps 6 --warmup 2 --parity-frames 3 > gpurun_out/r3o/bench_p3.log 2>&1; echo "bench rc=$?")
kill $MON
tail -1 gpurun_out/r3o/bench_p3.log | cut -c1-200; sort -n gpurun_out/r3o/rss.log | tail -2
