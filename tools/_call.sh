set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_c41; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_wres_$c -- python $R/tools/gemm_wres_bench.py > $OUT/wres_$c.log 2>&1
done
cd $R
python - <<'PY' > $OUT/pmc_traffic_wres.txt
import csv, glob, os, re, collections
out = os.environ.get("OUT", "gpurun_out/r6_c41")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join("gpurun_out/r6_c41", "pmc_wres_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void sampt::", "")
            if "wres" in n or "conv_f16x3" in n:
                acc[(n, r["Grid_Size"] if "Grid_Size" in r else "")].append(float(r["Counter_Value"]))
    for (n, g), v in sorted(acc.items()):
        print(f"{c:11s} {n:34s} grid {g:>9s}  n={len(v):4d}  mean {sum(v) / len(v) / 1e3:9.1f} MB (counter KB; x2 for FETCH_SIZE on gfx950)")
PY
rm -rf $OUT/pmc_*_SIZE
cat $OUT/pmc_traffic_wres.txt | cut -c1-170
