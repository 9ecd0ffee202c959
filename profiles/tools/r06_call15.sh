#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c15; mkdir -p $out; root=$(pwd)
timeout 1500 python profiles/tools/ab_run.py --workloads files mixed records text --reps 3 --steps 5 --out $out/ab_stable.json 2>$out/err.log | tee $out/ab.txt
for v in base stable; do BROTLIG_HIP_SO=$root/build/abv/lib_$v.so timeout 300 python profiles/phase_profile.py files 16 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['workload'], 'solo', round(d['solo_rounds'] / d['rounds'], 4), 'rounds', d['rounds'])
"; done | tee $out/solo.txt
tail -3 $out/err.log
