#!/bin/bash
# Round 6: team levels in two passes (BROTLIG_TUNE_SPLIT_TEAMS=1: runs with a period dividing 8 first, the rest with teams of its own): in-process A/B.
export TMPDIR=/tmp
out=gpurun_out/r06c27; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text records samples16 runs --reps 3 --steps 5 --out $out/ab_split_presence.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
