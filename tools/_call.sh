# scratch script of the current gpurun call: the chip split in space (CU-masked streams: encoder on 28 / 24 CUs per XCD, window rounds
# and decoder chains on the rest) — correctness of the split path, stream timelines, quick bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c4; mkdir -p $OUT; cd $R
SAMPT_SIDE_CUS=4 timeout 600 python -m pytest tests/test_gpu_modules.py -q -x -k "end_to_end or stream_of_clips or pipelined_decoder or graph_replay or ragged" > $OUT/pytest_split.log 2>&1
tail -3 $OUT/pytest_split.log
tl() { # side_cus mixer_wgs dec_split extra-env
  SAMPT_SIDE_CUS=$1 SAMPT_PIPS_MIXER_WGS=$2 timeout 200 python tools/forward_timeline.py --dec-split $3 2>&1 | grep -v amdgpu.ids > $OUT/timeline_side$1_w$2_s$3.log
  echo "side=$1 wgs=$2 split=$3: $(tail -1 $OUT/timeline_side$1_w$2_s$3.log)"
}
tl 0 32 0; tl 4 32 0; tl 4 32 2; tl 4 32 1; tl 8 64 2; tl 8 64 1; tl 6 32 2; tl 2 32 2
for cfg in "0 32 0" "4 32 2" "8 64 2"; do set -- $cfg
  SAMPT_SIDE_CUS=$1 SAMPT_PIPS_MIXER_WGS=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 10 --warmup 3 --dec-split $3 > $OUT/bench_side$1_w$2_s$3.log 2>&1
  tail -1 $OUT/bench_side$1_w$2_s$3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$1 wgs=$2 split=$3', d['value'], d['value_pipelined'], d.get('timeline'), d['parity']['mask_iou_min'] if 'parity' in d else None, d['parity']['pass'] if 'parity' in d else None)"
done
