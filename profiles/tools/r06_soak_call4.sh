#!/bin/bash
# Round 6, fourth soak, on the FINAL build (one-word runs up to 48 bytes; fresh seeds): the three kernel selections and the pair form pinned on three wavefronts
mkdir -p gpurun_out/r06_soak4
timeout 600 python profiles/tools/soak.py 610000 12800 3600 2400 1 > gpurun_out/r06_soak4/soak_one_wavefront.json 2> gpurun_out/r06_soak4/err1.log; echo "mode 1 rc=$?"; tail -c 320 gpurun_out/r06_soak4/soak_one_wavefront.json; echo
timeout 600 python profiles/tools/soak.py 620000 12800 3600 2400 0 > gpurun_out/r06_soak4/soak_auto.json 2> gpurun_out/r06_soak4/err0.log; echo "mode 0 rc=$?"; tail -c 320 gpurun_out/r06_soak4/soak_auto.json; echo
timeout 600 python profiles/tools/soak.py 630000 6400 7200 1600 2 > gpurun_out/r06_soak4/soak_two_wavefronts.json 2> gpurun_out/r06_soak4/err2.log; echo "mode 2 rc=$?"; tail -c 320 gpurun_out/r06_soak4/soak_two_wavefronts.json; echo
BROTLIG_SOAK_GRID=3 timeout 600 python profiles/tools/soak.py 640000 9600 3600 1600 1 > gpurun_out/r06_soak4/soak_pair3.json 2> gpurun_out/r06_soak4/err3.log; echo "pair3 rc=$?"; tail -c 320 gpurun_out/r06_soak4/soak_pair3.json; echo
timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -q -x -k "periodic_runs" 2>&1 | tail -2
