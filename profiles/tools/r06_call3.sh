#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c3; mkdir -p $out
timeout 600 python profiles/tools/sched_times.py mixed runs:1 mixed:16:64 mixed:16:256 2>$out/err.log | tee $out/sched_times.jsonl
timeout 600 python profiles/tools/ab_run.py --workloads mixed runs:1 mixed:16:64 mixed:16:8 bc3 --reps 3 --steps 10 --out $out/ab.json 2>>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
