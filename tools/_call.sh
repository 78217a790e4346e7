set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2; do echo "== shipped"; timeout 100 python tools/attn_bench.py 2>&1 | grep "window\|global"; for f in NOSCATTER NODMATAB; do echo "== debug build $f"; ATTN_BENCH_LIB=tools/_ab/libsampt_$f.so timeout 100 python tools/attn_bench.py 2>&1 | grep "window\|global"; done; done
