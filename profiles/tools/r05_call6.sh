#!/bin/bash
out=gpurun_out/r05c6; mkdir -p $out; root=$(pwd)
for v in bk2 base; do for w in mixed text records; do echo -n "$v "; BROTLIG_HIP_SO=$root/build/abv/lib_$v.so timeout 120 python profiles/phase_profile.py $w 16 2>> $out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['workload'], 'rounds', d['rounds'], 'solo_rounds', d['solo_rounds'], 'frac %.4f' % (d['solo_rounds'] / d['rounds']), 'halves_per_level', d['halves_per_level'], 'halves_per_group', d['halves_per_group'], 'tables', d['tables'], 'levels/round', d['levels_per_round'])
"; done; done | tee $out/solo_rounds.txt
