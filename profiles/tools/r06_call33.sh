#!/bin/bash
# Round 6: where the two-kernel launch loses time -- probes: 256 + b = the main kernel on the split prefix, no light launch; 512 + b = the main kernel as ever, light launch beside it
export TMPDIR=/tmp
out=gpurun_out/r06c33; mkdir -p $out
BROTLIG_ENABLE_DEBUG_KNOBS=1 timeout 900 python profiles/tools/light_ab.py --workloads text --buckets 296 552 --reps 3 --steps 5 --out $out/light_probe.json 2>$out/err.log | tee $out/ab.txt
tail -5 $out/err.log
