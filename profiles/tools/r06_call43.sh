#!/bin/bash
# Round 6: the small-texture batches on the library as it was before the no-unroll build and on the product, same box, alternating.
export TMPDIR=/tmp
out=gpurun_out/r06c43; mkdir -p $out
for rep in 1 2; do
  for lib in build/old/lib_r06_before.so brotli_g_sdk_amd/csrc/libbrotlig_hip.so; do
    echo "== $lib (rep $rep)"
    BROTLIG_HIP_SO=$(pwd)/$lib timeout 600 python profiles/tools/r05_many_textures.py 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-90s step %.4f decode %.4f rest %.4f  %.1f GB/s' % (d['batch'][:90], d['step_ms'], d['decode_kernel_ms'], d['prepare_and_decondition_ms'], d['GBps_step']))"
  done
done | tee $out/many_textures_ab.txt
tail -3 $out/err.log
