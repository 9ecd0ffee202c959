#!/bin/bash
# Round 5: batches between one and two pages per wavefront taken FOLDED (lib_new.so) against the commit before (lib_base.so): the many-textures
# batches and the many-streams batches one library after the other on one box, the benchmark's mixed workload interleaved, then the new device test
mkdir -p gpurun_out/r05_fold
for so in build/abv/lib_base.so build/abv/lib_new.so; do
  echo "== $so"; BROTLIG_HIP_SO=$(pwd)/$so timeout 120 python profiles/tools/r05_many_textures.py 2>/dev/null | tee gpurun_out/r05_fold/many_textures_$(basename $so .so).jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['textures'], d['step_ms'], d['decode_kernel_ms'], d['GBps_step'], d['batch'][:40])
"
done 2>&1 | tee gpurun_out/r05_fold/many_textures.txt
python profiles/tools/ab_run.py --workloads mixed --reps 3 --steps 5 --out gpurun_out/r05_fold/ab_mixed.json 2>&1 | tail -3
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
