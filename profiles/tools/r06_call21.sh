#!/bin/bash
# Round 6: the machine scheduler's other strategies for the decode kernel (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause): in-process A/B.
export TMPDIR=/tmp
out=gpurun_out/r06c21; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text files records bc3 --reps 3 --steps 5 --out $out/ab_sched.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
