# scratch script of the current gpurun call: the ViT neck's 3 x 3 convolution on the halo kernel (LayerNorm2d writes fp16 planes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c40; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_bench_parity.py -x -q -k "vit or layernorm or sam or golden or predictor or parity or smoke" > $OUT/pytest_vit.log 2>&1; tail -3 $OUT/pytest_vit.log | cut -c1-300
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $OUT/prof -o clip -- python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 4 --warmup 2 > $OUT/rocprof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python $R/tools/rocprof_by_grid.py "$DB" "" 6 > $OUT/clip_by_grid.txt 2>&1; rm -rf $OUT/prof
grep "k_conv_f16x3<\|halo_x3<128\|layernorm_rows_v4<1>" $OUT/clip_by_grid.txt | cut -c1-140
grep "^{" $OUT/rocprof.log | python -c "
import json,sys
for l in sys.stdin: d=json.loads(l); print(d['value'], d.get('timeline'))"
cd $R; timeout 600 python bench.py --steps 4 --warmup 2 --no-secondary --no-roofline --no-pipelined > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); p=d["parity"]; print(d["value"], d.get("timeline"), p["pass"], p["mask_iou_min"], p["logit_max_abs"])
PY
