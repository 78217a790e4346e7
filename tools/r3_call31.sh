set -x
mkdir -p gpurun_out/r3zz
cd /root/repo
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline"
for w in 28 27 29 28,28,32 26,28,32 28; do
  SAMPT_ENC_WGS=$w timeout 200 python bench.py $B > gpurun_out/r3zz/bench_wgs_$w.log 2>&1; echo "wgs $w: $(tail -1 gpurun_out/r3zz/bench_wgs_$w.log | grep -o '"value": [0-9.]*')"
done
