#!/bin/bash
# Round 6, on the final build: the distance ring skipped when no lane refers to it (BROTLIG_TUNE_RING_SKIP=1), the own-lane / team thresholds, GVN PRE off once more.
export TMPDIR=/tmp
out=gpurun_out/r06c58; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text records samples16 files runs --reps 3 --steps 5 --out $out/ab_pair_instantiation_alone.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
