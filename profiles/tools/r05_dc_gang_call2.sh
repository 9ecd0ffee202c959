#!/bin/bash
# Round 5: the adopted de-conditioning distribution (gang size by the batch: lib_new.so) against the commit before (lib_base.so), then the GPU suite
mkdir -p gpurun_out/r05_dc_gang
python profiles/tools/ab_run.py --workloads bc3 --reps 3 --steps 5 --out gpurun_out/r05_dc_gang/ab_bc3_adopted.json 2>&1 | tail -4
for so in build/abv/lib_base.so build/abv/lib_new.so; do
  echo "== $so"; BROTLIG_HIP_SO=$(pwd)/$so timeout 120 python profiles/tools/r05_many_textures.py 2>/dev/null | tee gpurun_out/r05_dc_gang/many_textures_$(basename $so .so).jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['textures'], d['step_ms'], d['prepare_and_decondition_ms'], d['GBps_step'], d['batch'][:40])
"
done 2>&1 | tee gpurun_out/r05_dc_gang/many_textures_adopted.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
