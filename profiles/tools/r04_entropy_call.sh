#!/bin/bash
# the entropy decode alone (experimental split build, BROTLIG_SPLIT=1) on the literal-/copy-heavy classes.
root=$(pwd); out=$root/gpurun_out/r04entropy; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
for w in mixed samples16 bc3 text; do
  extra=""; [ $w = bc3 ] && extra="--streams 256"
  BROTLIG_SPLIT=1 BROTLIG_HIP_SO=$root/build/abv/lib_split.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -o f -- \
     python $root/bench.py --workload $w $extra --steps 3 --warmup 1 --no-cpu-baseline --no-alt-parse > $out/$w.log 2>&1
  python - $out/$w $w <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "brotlig" in r["Name"] and float(r["Percentage"]) > 1:
            print(sys.argv[2], r["Name"][:60], "avg_ms", round(float(r["AverageNs"]) / 1e6, 3), "calls", r["Calls"])
PY
  grep -o '"value": [0-9.]*\|"bit_exact": [a-z]*' $out/$w.log | tr '\n' ' '; echo
done
cd $root; find $out -name '*.csv' -size +1M -delete; find $out -name '*agent_info*' -delete
python profiles/tools/latency.py > $out/latency.json 2>/dev/null; cat $out/latency.json
