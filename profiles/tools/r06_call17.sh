#!/bin/bash
# Round 6, end-game priority, second pass: 1 / 2 / 3 eighths of the final generation at s_setprio 3, and two tiers (the eighths in front held at 2).
export TMPDIR=/tmp
out=gpurun_out/r06c17; mkdir -p $out
timeout 1200 python profiles/tools/ab_run.py --workloads mixed text files records --reps 3 --steps 5 --out $out/ab_endgame2.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
