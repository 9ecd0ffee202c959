#!/bin/bash
# Product library at several grid sizes (workgroups per CU; more than fit simply queue behind the resident ones).
out=gpurun_out/occ_sweep; mkdir -p $out
for n in ${NS:-12 13 14 16}; do
  for w in ${WL:-mixed text}; do
  BROTLIG_WG_PER_CU=$n python bench.py --workload $w --no-cpu-baseline --no-alt-parse --steps 4 --warmup 1 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wg_per_cu $n', '$w', d['value'], 'GB/s kernel_ms', d['roofline']['kernel_ms'], 'exact', d['bit_exact'])
" | tee -a $out/summary.txt
  done
done
