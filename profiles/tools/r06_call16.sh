#!/bin/bash
# Round 6, end-game priority (BROTLIG_TUNE_ENDGAME = 2 / 4 / 8 eighths of the final generation at s_setprio 3): in-process A/B on mixed, text, files,
# and the wavefronts' leaving times per variant.
export TMPDIR=/tmp
out=gpurun_out/r06c16; mkdir -p $out
timeout 900 python profiles/tools/ab_run.py --workloads mixed text files --reps 3 --steps 5 --out $out/ab_endgame.json 2>$out/err.log | tee $out/ab.txt
for v in base eg2 eg4 eg8; do
  for w in mixed text; do
    echo -n "$v " >> $out/wave_times.txt
    BROTLIG_HIP_SO=$(pwd)/build/abv/lib_$v.so timeout 300 python profiles/tools/wave_times.py --workload $w 2>>$out/err.log | tee -a $out/wave_times.txt
  done
done
tail -3 $out/err.log
