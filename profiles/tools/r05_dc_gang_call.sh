#!/bin/bash
# Round 5: the de-conditioning kernel's work distribution -- the old one (streams over blockIdx.y: lib_base.so = the commit before) against
# gangs of 1 / 8 / 32 / 64 / 256 wavefronts over one list of super-tiles -- on config 4 (ab_run, interleaved) and on the batches of many
# small textures (r05_many_textures.py, one library after the other on the same box).
mkdir -p gpurun_out/r05_dc_gang
python profiles/tools/ab_run.py --workloads bc3 --reps 3 --steps 5 --out gpurun_out/r05_dc_gang/ab_bc3.json 2>&1 | tail -12
for so in build/abv/lib_base.so build/abv/lib_g1.so build/abv/lib_g8.so build/abv/lib_g32.so build/abv/lib_g64.so build/abv/lib_g256.so; do
  echo "== $so"; BROTLIG_HIP_SO=$(pwd)/$so timeout 120 python profiles/tools/r05_many_textures.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['textures'], d['step_ms'], d['prepare_and_decondition_ms'], d['batch'][:40])
"
done 2>&1 | tee gpurun_out/r05_dc_gang/many_textures_by_variant.txt
