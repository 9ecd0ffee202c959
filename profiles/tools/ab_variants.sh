#!/bin/bash
# A/B builds of the decode kernel with tunables overridden, compared on one box in one gpurun call:
#   bash profiles/tools/ab_variants.sh build "NAME:-DFLAG=.. -DFLAG=.." ...     (here: hipcc cross-compiles)
#   bash profiles/tools/ab_variants.sh run [workloads]                           (on the MI355X box)
if [ "$1" = build ]; then
  shift; rm -rf build/abv; mkdir -p build/abv
  for v in "base:" "$@"; do
    name=${v%%:*}; flags=${v#*:}
    ( cd brotli_g_sdk_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-unroll-loops -fPIC -shared $flags -I ../../include -I . \
        -o ../../build/abv/lib_$name.so brotlig_hip.hip brotlig_streamer.hip 2>&1 | grep -i " error" ) &
  done
  wait; ls build/abv
else
  shift; out=gpurun_out/abv; mkdir -p $out
  for w in ${@:-mixed}; do for so in build/abv/lib_base.so build/abv/lib_*.so; do
    BROTLIG_HIP_SO=$(pwd)/$so python bench.py --workload $w --no-cpu-baseline --no-alt-parse --steps 5 --warmup 2 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$so', '$w', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], d['bit_exact'])
" | tee -a $out/summary.txt
  done; done
fi
