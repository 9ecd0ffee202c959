#!/bin/bash
# Round 6: the table build's fifteen-step loops unrolled by pragma (BROTLIG_TUNE_TABLE_UNROLL=1) and dc_init's constant loops; `old` = the library before the no-unroll build.
export TMPDIR=/tmp
out=gpurun_out/r06c45; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text bc3 --reps 3 --steps 5 --out $out/ab_delta_unroll.json 2>$out/err.log | tee $out/ab.txt
for lib in build/abv/lib_old.so build/abv/lib_base.so build/abv/lib_deltaun.so; do
  echo "== $lib"
  BROTLIG_HIP_SO=$(pwd)/$lib timeout 600 python profiles/tools/r05_many_textures.py 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-90s step %.4f decode %.4f rest %.4f  %.1f GB/s' % (d['batch'][:90], d['step_ms'], d['decode_kernel_ms'], d['prepare_and_decondition_ms'], d['GBps_step']))"
done | tee $out/many_textures_ab.txt
tail -3 $out/err.log
