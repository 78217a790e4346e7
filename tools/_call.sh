# scratch script of the current gpurun call: x3 mixer with the 4-slot ring / nt loads / 16-slab batches (kernel tests + chain alone),
# encoder batch size x GEMM workgroups per XCD (tile-count quantisation: 12 frames x 30 workgroups gives whole rounds)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c8; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "pips_mix" > $OUT/pytest_kernels.log 2>&1; tail -2 $OUT/pytest_kernels.log
for cfg in "2 16" "2 32"; do set -- $cfg
  SAMPT_PIPS_MIXER=$1 SAMPT_PIPS_MIXER_WGS=$2 timeout 200 python tools/tracker_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/tracker_bench_m$1_w$2.log
  echo "mixer=$1 wgs=$2: $(tail -1 $OUT/tracker_bench_m$1_w$2.log)"
done
export SAMPT_PIPS_MIXER=2 SAMPT_PIPS_MIXER_WGS=16
for cfg in "8 30" "12 30" "24 30" "12 28" "12 31" "6 30" "12 30/30/30/32"; do set -- $cfg
  SAMPT_ENC_WGS=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 8 --warmup 3 --encode-batch $1 > "$OUT/bench_b$1_e${2//\//-}.log" 2>&1
  tail -1 "$OUT/bench_b$1_e${2//\//-}.log" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch=$1 enc=$2', d['value'], d.get('timeline'))"
done
