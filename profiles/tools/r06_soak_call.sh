#!/bin/bash
# Round 6 soak on the schedule kernel + job records: the three kernel selections on fresh seeds, and the two-pages-per-wavefront form pinned on three wavefronts
mkdir -p gpurun_out/r06_soak
timeout 300 python profiles/tools/soak.py 360000 6400 1800 1200 1 > gpurun_out/r06_soak/soak_one_wavefront.json 2> gpurun_out/r06_soak/err1.log; echo "mode 1 rc=$?"; tail -c 320 gpurun_out/r06_soak/soak_one_wavefront.json; echo
timeout 300 python profiles/tools/soak.py 370000 6400 1800 1200 0 > gpurun_out/r06_soak/soak_auto.json 2> gpurun_out/r06_soak/err0.log; echo "mode 0 rc=$?"; tail -c 320 gpurun_out/r06_soak/soak_auto.json; echo
timeout 300 python profiles/tools/soak.py 380000 3200 3600 800 2 > gpurun_out/r06_soak/soak_two_wavefronts.json 2> gpurun_out/r06_soak/err2.log; echo "mode 2 rc=$?"; tail -c 320 gpurun_out/r06_soak/soak_two_wavefronts.json; echo
BROTLIG_SOAK_GRID=3 timeout 300 python profiles/tools/soak.py 390000 4800 1800 800 1 > gpurun_out/r06_soak/soak_pair3.json 2> gpurun_out/r06_soak/err3.log; echo "pair3 rc=$?"; tail -c 320 gpurun_out/r06_soak/soak_pair3.json; echo
