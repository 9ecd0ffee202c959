#!/bin/bash
# Ablation builds of the decode kernel (compile here, run on the MI355X box through gpurun):
#   bash profiles/tools/ablate.sh build            -> build/ablate/libbrotlig_hip_<mask>.so
#   bash profiles/tools/ablate.sh run <tag> [workload]   (on the GPU box)
set -u
masks="0 1 2 4 8 32 64 128 239"
if [ "$1" = build ]; then
  mkdir -p build/ablate
  for m in $masks; do
    ( cd brotli_g_sdk_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-unroll-loops -fPIC -shared -DBROTLIG_ABLATE=$m -I ../../include -I . \
        -o ../../build/ablate/libbrotlig_hip_$m.so brotlig_hip.hip brotlig_streamer.hip ) &
  done
  wait; ls -la build/ablate/*.so
else
  tag=$2; wl=${3:-mixed}; out=gpurun_out/$tag; mkdir -p $out
  for m in $masks; do
    BROTLIG_HIP_SO=$(pwd)/build/ablate/libbrotlig_hip_$m.so python bench.py --workload $wl --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('mask $m', '$wl', 'kernel_ms', d['roofline']['kernel_ms'], 'exact', d['bit_exact'])
" | tee -a $out/ablate_$wl.txt
  done
fi
