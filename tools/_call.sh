# scratch script of the current gpurun call: split-fp16 mixer on CU-masked side CUs (2 / 4 per XCD), decoder chains on the whole chip
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c6; mkdir -p $OUT; cd $R
tl() { # side_cus mixer wgs dec_split
  SAMPT_SIDE_CUS=$1 SAMPT_PIPS_MIXER=$2 SAMPT_PIPS_MIXER_WGS=$3 timeout 200 python tools/forward_timeline.py --dec-split $4 2>&1 | grep -v amdgpu.ids > $OUT/timeline_side$1_m$2_w$3_s$4.log
  echo "side=$1 mixer=$2 wgs=$3 split=$4: $(tail -1 $OUT/timeline_side$1_m$2_w$3_s$4.log)"
}
tl 2 2 16 0; tl 2 2 16 2; tl 2 2 16 1; tl 4 2 32 2; tl 4 2 32 1; tl 3 2 16 1; tl 1 2 16 1; tl 4 2 16 1; tl 2 1 32 1
