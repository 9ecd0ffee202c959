#!/bin/bash
# Pairing policy pinned (BROTLIG_POLICY = quarters of a page within which a free half waits for its neighbour).
out=gpurun_out/policy; mkdir -p $out
for pol in ${POLS:-0 1 2 4}; do
  for w in ${WL:-mixed text}; do
  BROTLIG_POLICY=$pol python bench.py --workload $w --no-cpu-baseline --no-alt-parse --steps 4 --warmup 1 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('policy $pol', '$w', d['value'], 'GB/s kernel_ms', d['roofline']['kernel_ms'], 'exact', d['bit_exact'])
" | tee -a $out/summary.txt
  done
done
