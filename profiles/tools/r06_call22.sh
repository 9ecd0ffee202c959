#!/bin/bash
# Round 6: branch-weight hints on the symbol decode's rare paths (BROTLIG_TUNE_EXPECT=1: the long-code route and the overflow slot out of line): in-process A/B.
export TMPDIR=/tmp
out=gpurun_out/r06c22; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text files records samples16 bc3 --reps 3 --steps 5 --out $out/ab_expect.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
