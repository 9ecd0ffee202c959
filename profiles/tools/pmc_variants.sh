#!/bin/bash
# SQ counters of the decode kernel for several builds of build/abv/ (one rocprofv3 --pmc pass each, kernel-trace only):
#   bash profiles/tools/pmc_variants.sh <tag> <variant> [<variant> ...]        (WORKLOAD=mixed by default)
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
set1="${PMC_SET1:-SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY}"
set2="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"
for v in "$@"; do
  for s in 1 2; do
    eval set=\$set$s
    [ "$s" = 2 ] && [ -z "$PMC_SET2" ] && continue
    BROTLIG_HIP_SO=$root/build/abv/lib_$v.so timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${v}_$s -o f -- \
      python $root/bench.py --workload ${WORKLOAD:-mixed} --steps 1 --warmup 1 --no-cpu-baseline --no-alt-parse > $out/${v}_$s.log 2>&1 || echo "$v pass $s failed: $(tail -2 $out/${v}_$s.log)"
  done
done
cd $root
python - "$out" "$@" <<'PY'
import csv, glob, collections, sys, json
out, names = sys.argv[1], sys.argv[2:]
table = {}
for v in names:
    acc = collections.defaultdict(list)
    for p in glob.glob(f"{out}/{v}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "brotlig_decode_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    table[v] = {k: sum(x) / len(x) for k, x in acc.items()}
keys = sorted({k for t in table.values() for k in t})
print("%-24s" % "counter" + "".join("%16s" % v[:15] for v in names))
for k in keys:
    print("%-24s" % k + "".join("%16.4g" % table[v].get(k, float("nan")) for v in names))
json.dump(table, open(f"{out}/summary.json", "w"), indent=1)
PY
find $out -name '*.csv' -size +2M -delete; find $out -name '*agent_info*' -delete
