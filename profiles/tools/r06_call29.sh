#!/bin/bash
# Round 6: runs first in a team level (BROTLIG_TUNE_RUNS_FIRST=1: a pre-pass for the pieces that repeat a period of 1, 2, 4 or 8 bytes, then the round-4 team code for the rest): in-process A/B.
export TMPDIR=/tmp
out=gpurun_out/r06c29; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text records samples16 runs runs:1 bc3 files --reps 3 --steps 5 --out $out/ab_runs_first_z.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
