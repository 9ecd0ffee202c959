#!/bin/bash
# Round 6, traffic experiment: the bit readers' 8 bytes in flight as 16 bytes in an LDS slot filled by the direct global -> LDS load (BROTLIG_EXP_GLDS=1),
# against the register reader with the same LDS footprint (=2): bit-exactness and time on the full kernel, TCC counters on the ablated one (abl255).
export TMPDIR=/tmp
out=gpurun_out/r06c13; mkdir -p $out; root=$(pwd)
timeout 900 python profiles/tools/ab_run.py --workloads mixed text files --reps 2 --steps 5 --out $out/ab_glds.json 2>$out/err.log | tee $out/ab.txt
cd /tmp
for v in 1 2; do
  BROTLIG_HIP_SO=$root/build/abl/lib_abl255_exp$v.so timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $root/$out/tcc_abl255_exp$v -o f -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > $root/$out/tcc_abl255_exp$v.log 2>&1
  BROTLIG_HIP_SO=$root/build/abv/lib_$( [ $v = 1 ] && echo glds || echo ctrl ).so timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $root/$out/tcc_full_exp$v -o f -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > $root/$out/tcc_full_exp$v.log 2>&1
done
cd $root
python - <<'PY'
import csv, glob, collections, json
res = {}
for tag in ("tcc_abl255_exp1", "tcc_abl255_exp2", "tcc_full_exp1", "tcc_full_exp2"):
    acc = collections.defaultdict(list); dur = []
    for p in glob.glob(f"gpurun_out/r06c13/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "brotlig_decode_kernel(" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    res[tag] = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
    if dur: res[tag]["kernel_ms_under_rocprof"] = round(sum(dur) / len(dur), 3)
    print(tag, {k: ("%.4g" % v) for k, v in res[tag].items()})
json.dump(res, open("gpurun_out/r06c13/tcc_summary.json", "w"), indent=1)
PY
find $out -name '*.csv' -size +4M -delete; find $out -name '*agent_info*' -delete
tail -3 $out/err.log
