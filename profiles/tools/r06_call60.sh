#!/bin/bash
# Round 6 as a whole, in one process on one box: the library as round 5 left it (commit 1b885a0, its own flags), the library after this round's schedule work
# (before the no-unroll build), and the final build (`base`).  `step` is the figure to compare across them: the round-5 library launches five kernels in front of the decode.
export TMPDIR=/tmp
out=gpurun_out/r06c60; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text records samples16 files runs bc3 --reps 3 --steps 5 --out $out/ab_prio_and_layout.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
