#!/bin/bash
# SQ counter passes on the default bench workload (each pass its own rocprofv3 run, kernel-trace only).
#   bash profiles/tools/pmc_quick.sh <tag>
tag=${1:-pmcq}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_I8"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o f -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $out/p$i.log)"
done
cd $root
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for p in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "brotlig_decode_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/summary.txt", "w") as f:
    for k in sorted(acc):
        line = "%-28s %.4g" % (k, sum(acc[k]) / len(acc[k]))
        print(line); f.write(line + "\n")
PY
find $out -name '*.csv' -size +4M -delete; find $out -name '*agent_info*' -delete
