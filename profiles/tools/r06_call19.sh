#!/bin/bash
# Round 6: the piece-start bitmap set before the literal decode and looked up after it (BROTLIG_TUNE_LATE_DEPS = 1 / 2): in-process A/B.
export TMPDIR=/tmp
out=gpurun_out/r06c19; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text files records samples16 runs bc3 --reps 3 --steps 5 --out $out/ab_late.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
