#!/bin/bash
# Round 6, final build: what the page starts alone cost (-DBROTLIG_ABLATE=256: job fetch, bit readers, the three table builds, no round), the entropy decode without the
# assembly (255) and the kernel without copy levels (1) -- wrong output by construction, timing only.
export TMPDIR=/tmp
out=gpurun_out/r06c61; mkdir -p $out
for n in base abl256 abl255 abl1; do
  for w in mixed text; do
    BROTLIG_HIP_SO=$(pwd)/build/abv/lib_$n.so python bench.py --workload $w --no-cpu-baseline --no-alt-parse --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$n', '$w', 'kernel_ms', d['roofline']['kernel_ms'], 'exact', d['bit_exact'])"
  done
done | tee $out/ablate_final.txt
