# scratch script of the current gpurun call: kernel traces of the window chain alone (fused mixer at 32 / 64 workgroups, four-launch
# blocks) and of the blocking clip, launch-ordered excerpt of one iteration
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c2; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "pips_mix" > $OUT/pytest_kernels.log 2>&1
tail -3 $OUT/pytest_kernels.log
cd /tmp && export TMPDIR=/tmp
for cfg in "1 32" "1 64" "0 32"; do set -- $cfg
  SAMPT_PIPS_MIXER=$1 SAMPT_PIPS_MIXER_WGS=$2 timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_$1_$2 -o trk -- python $R/tools/tracker_bench.py > $OUT/rocprof_$1_$2.log 2>&1
  DB=$(find $OUT/prof_$1_$2 -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py "$DB" > $OUT/tracker_kernel_stats_m$1_w$2.txt 2>&1
  [ -n "$DB" ] && python $R/tools/rocprof_sequence.py "$DB" 140 > $OUT/tracker_sequence_m$1_w$2.txt 2>&1
  rm -rf $OUT/prof_$1_$2
  head -12 $OUT/tracker_kernel_stats_m$1_w$2.txt
done
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_clip -o clip -- python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline --no-pipelined --steps 4 --warmup 2 > $OUT/rocprof_clip.log 2>&1
DB=$(find $OUT/prof_clip -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py "$DB" 168 > $OUT/clip_kernel_stats.txt 2>&1
rm -rf $OUT/prof_clip
head -30 $OUT/clip_kernel_stats.txt
