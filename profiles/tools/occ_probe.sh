#!/bin/bash
# Occupancy probe: the product kernel with extra dynamic LDS so that only N workgroups fit a CU (LDS is allocated in
# 1280-byte granules, 128 per CU).  build/occ/libocc.so = the product sources with BROTLIG_WG_PER_CU / BROTLIG_DYN_LDS.
out=gpurun_out/occ; mkdir -p $out
for cfg in "12 0" "11 1312" "10 2592" "9 4000" "8 7712" "6 14112"; do
  set -- $cfg
  for w in ${WL:-mixed text}; do
  BROTLIG_WG_PER_CU=$1 BROTLIG_DYN_LDS=$2 BROTLIG_HIP_SO=$(pwd)/build/occ/libocc.so python bench.py --workload $w --no-cpu-baseline --no-alt-parse --steps 4 --warmup 1 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wg_per_cu $1 dyn_lds $2', '$w', d['value'], 'GB/s kernel_ms', d['roofline']['kernel_ms'], 'exact', d['bit_exact'])
" | tee -a $out/summary.txt
  done
done
