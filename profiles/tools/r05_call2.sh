#!/bin/bash
# Round 5, second call: pairing-policy sweep and de-conditioning grid sweep on the current kernels, the launch's tail on the PRODUCT kernel
# (wave-times build), the L2 / fabric counters of the builds without far sources, a kernel trace of config 4.
out=gpurun_out/r05c2; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
POLS="0 1 2 4" WL="mixed" bash profiles/tools/policy_sweep.sh > $out/policy.txt 2>&1; cat $out/policy.txt
for n in 16 24 32 48 64; do
  BROTLIG_DC_PER_CU=$n python bench.py --workload bc3 --streams 256 --no-cpu-baseline --no-alt-parse --steps 5 --warmup 2 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('dc_per_cu $n', d['value'], 'GB/s step_ms', d['ms_per_step'], 'exact', d['bit_exact'])
" | tee -a $out/dc_sweep.txt
done
for w in mixed text; do
  BROTLIG_HIP_SO=$root/build/abv/lib_wavetimes.so timeout 120 python profiles/tools/wave_times.py --workload $w 2>>$out/err.log | tee $out/wave_times_$w.json
done
cd /tmp
for v in base abl8 abl255; do
  BROTLIG_HIP_SO=$root/build/abv/lib_$v.so timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $root/$out/tcc_$v -o f -- \
    python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > $root/$out/tcc_$v.log 2>&1 || echo "$v failed: $(tail -2 $root/$out/tcc_$v.log)"
done
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace_bc3 -o t -- python $root/bench.py --workload bc3 --streams 256 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-parse > $root/$out/trace_bc3.log 2>&1
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
find $out -name '*_kernel_trace.csv' -size +4M -delete; find $out -name '*agent_info*' -delete; find $out -name '*counter_collection.csv' -size +2M -delete
cat $out/trace_bc3/*kernel_stats.csv 2>/dev/null | head -4 | cut -c1-160
