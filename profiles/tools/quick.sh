#!/bin/bash
# Quick perf check on the MI355X box: bench lines for the given workloads + the phase profile.
#   bash profiles/tools/quick.sh <tag> [workloads...]      (PROFILE="mixed text" chooses the phase profiles)
tag=${1:-quick}; shift
wl=${@:-"mixed text runs bc3 samples16 records"}
out=gpurun_out/$tag; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { echo SMOKE FAILED; tail -5 $out/smoke.log; }
for w in $wl; do
  extra=""; [ $w = bc3 ] && extra="--streams 256"
  python bench.py --workload $w $extra --no-cpu-baseline --steps 5 --warmup 2 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', d['value'], 'GB/s  kernel_ms', d['roofline']['kernel_ms'], 'exact', d['bit_exact'])
" | tee -a $out/summary.txt
done
for k in ${PROFILE:-mixed}; do python profiles/phase_profile.py $k 16; done > $out/phase_profile.jsonl 2>>$out/err.log
cat $out/phase_profile.jsonl
