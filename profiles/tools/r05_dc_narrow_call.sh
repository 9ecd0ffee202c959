#!/bin/bash
# Round 5: narrow mips as super-tiles of several row pairs (lib_new.so) against the commit before (lib_base.so): config 4 interleaved, the batches of
# many small textures one library after the other, then the GPU suite and a short soak with pre-conditioned streams in the majority
mkdir -p gpurun_out/r05_dc_narrow
python profiles/tools/ab_run.py --workloads bc3 --reps 3 --steps 5 --out gpurun_out/r05_dc_narrow/ab_bc3.json 2>&1 | tail -4
for so in build/abv/lib_base.so build/abv/lib_new.so; do
  echo "== $so"; BROTLIG_HIP_SO=$(pwd)/$so timeout 120 python profiles/tools/r05_many_textures.py 2>/dev/null | tee gpurun_out/r05_dc_narrow/many_textures_$(basename $so .so).jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['textures'], d['step_ms'], d['prepare_and_decondition_ms'], d['GBps_step'], d['batch'][:40])
"
done 2>&1 | tee gpurun_out/r05_dc_narrow/many_textures.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 200 python profiles/tools/soak.py 340000 800 4800 400 0 2> gpurun_out/r05_dc_narrow/soak_err.log | tee gpurun_out/r05_dc_narrow/soak.json | tail -c 300; echo
