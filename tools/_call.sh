# scratch script of the current gpurun call (overwritten per call; the logs it leaves are copied to profiles/r4_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_v4; mkdir -p $OUT; cd $R
P="--steps 10 --warmup 3 --no-secondary --no-roofline --parity-frames 4"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['parity']; print('$2', d['value'], d['value_per_forward'], p['pass'], p['mask_iou_min'], p['masks_compared'], p['traj_index_identical'], p['traj_index_differing'], p['vis_identical'])"; }
timeout 400 python bench.py $P --model vit_b > $OUT/bench_cfg2_vitb.log 2>&1; show $OUT/bench_cfg2_vitb.log cfg2
timeout 400 python bench.py $P --tracker cotracker --neg-points 8 --frames 50 --cotracker-delta-scale 0.001 > $OUT/bench_cfg3_cotracker_T50_conditioned.log 2>&1; show $OUT/bench_cfg3_cotracker_T50_conditioned.log cfg3
timeout 400 python bench.py $P --objects 3 > $OUT/bench_cfg4_3obj.log 2>&1; show $OUT/bench_cfg4_3obj.log cfg4
timeout 600 python bench.py --steps 4 --warmup 2 --no-secondary --no-roofline --parity-frames 4 --hq --tracker cotracker --square 1024 --points 16 --objects 5 --frames 64 --cotracker-delta-scale 0.001 > $OUT/bench_cfg5_hq_T64.log 2>&1; show $OUT/bench_cfg5_hq_T64.log cfg5
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o vith -- python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 5 --warmup 2 > $OUT/rocprof_f16.log 2>&1
cd $R; python tools/rocprof_summary.py $(find $OUT/prof -name "*.db" | head -1) 312 > $OUT/vith_kernel_stats.txt 2>&1; head -12 $OUT/vith_kernel_stats.txt; rm -rf $OUT/prof
