#!/bin/bash
# Round 5, third call: schedule buckets of a half / quarter / eighth octave (tiled and untiled pages), the L2 / fabric counters of the builds
# without far sources, the producer-side dependency pass of the two-wavefront kernel (unfinished in round 4).
out=gpurun_out/r05c3; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
timeout 500 python profiles/tools/ab_run.py --workloads mixed text records samples16 --reps 3 --steps 5 --out $out/ab_buckets.json 2> $out/ab.err | tee $out/ab_buckets.txt
( time timeout 600 python profiles/tools/ab_run.py --workloads mixed --distinct 4096 --reps 3 --steps 5 --out $out/ab_buckets_distinct4096.json ) 2>> $out/ab.err | tee $out/ab_buckets_distinct4096.txt
tail -4 $out/ab.err
cd /tmp
for v in abl8 abl255; do
  BROTLIG_HIP_SO=$root/build/abl/lib_$v.so timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $root/$out/tcc_$v -o f -- \
    python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > $root/$out/tcc_$v.log 2>&1 || echo "$v failed: $(tail -2 $root/$out/tcc_$v.log)"
done
BROTLIG_HIP_SO=$root/build/abv/lib_base.so timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $root/$out/tcc_base -o f -- \
    python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > $root/$out/tcc_base.log 2>&1 || echo "base failed"
cd $root
python - $out <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
table = {}
for v in ("base", "abl8", "abl255"):
    acc = collections.defaultdict(list)
    for p in glob.glob(f"{out}/tcc_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "brotlig_decode_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    table[v] = {k: sum(x) / len(x) for k, x in acc.items()}
json.dump(table, open(f"{out}/tcc_summary.json", "w"), indent=1)
print(json.dumps(table, indent=1))
PY
for v in base duo1 duo2; do
  for m in two_wavefronts; do BROTLIG_HIP_SO=$root/build/duo/lib_$v.so timeout 100 python profiles/tools/page_latency.py $m 2>> $out/lat.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['class'], 'kernel_ms', d['kernel_ms'])
"; done
done | tee $out/duo_deps_page_latency.txt
for v in base duo1 duo2; do BROTLIG_HIP_SO=$root/build/duo/lib_$v.so timeout 120 python profiles/tools/latency.py two_wavefronts 2>> $out/lat.err | cut -c1-900 | sed "s/^/$v /"; done | tee $out/duo_deps_latency.txt
find $out -name '*_kernel_trace.csv' -size +4M -delete; find $out -name '*agent_info*' -delete; find $out -name '*counter_collection.csv' -size +2M -delete
