#!/bin/bash
# Round 6, second soak on the final build (fresh seeds, twice the streams): the three kernel selections and the pair form pinned on three wavefronts
mkdir -p gpurun_out/r06_soak2
timeout 600 python profiles/tools/soak.py 460000 12800 3600 2400 1 > gpurun_out/r06_soak2/soak_one_wavefront.json 2> gpurun_out/r06_soak2/err1.log; echo "mode 1 rc=$?"; tail -c 320 gpurun_out/r06_soak2/soak_one_wavefront.json; echo
timeout 600 python profiles/tools/soak.py 470000 12800 3600 2400 0 > gpurun_out/r06_soak2/soak_auto.json 2> gpurun_out/r06_soak2/err0.log; echo "mode 0 rc=$?"; tail -c 320 gpurun_out/r06_soak2/soak_auto.json; echo
timeout 600 python profiles/tools/soak.py 480000 6400 7200 1600 2 > gpurun_out/r06_soak2/soak_two_wavefronts.json 2> gpurun_out/r06_soak2/err2.log; echo "mode 2 rc=$?"; tail -c 320 gpurun_out/r06_soak2/soak_two_wavefronts.json; echo
BROTLIG_SOAK_GRID=3 timeout 600 python profiles/tools/soak.py 490000 9600 3600 1600 1 > gpurun_out/r06_soak2/soak_pair3.json 2> gpurun_out/r06_soak2/err3.log; echo "pair3 rc=$?"; tail -c 320 gpurun_out/r06_soak2/soak_pair3.json; echo
