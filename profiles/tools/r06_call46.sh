#!/bin/bash
# Round 6: is the "presence cost" of code in the round loop a matter of where the code lands?  Block alignment (-mllvm -align-all-nofallthru-blocks) on the product and on the runs-first build.
export TMPDIR=/tmp
out=gpurun_out/r06c46; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text records samples16 files --reps 3 --steps 5 --out $out/ab_compiler_flags_3.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
