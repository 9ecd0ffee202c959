#!/bin/bash
# after the prepare kernels went to one workgroup per 64 streams and the de-conditioning kernel to one list of super-tiles: the soak on fresh seeds,
# pre-conditioned streams in the majority, in both kernel selections
mkdir -p gpurun_out/r05_soak2
timeout 260 python profiles/tools/soak.py 320000 2400 4800 800 0 > gpurun_out/r05_soak2/soak_auto.json 2> gpurun_out/r05_soak2/err0.log; echo "mode 0 rc=$?"; tail -c 300 gpurun_out/r05_soak2/soak_auto.json; echo
timeout 260 python profiles/tools/soak.py 330000 1600 4000 800 1 > gpurun_out/r05_soak2/soak_one_wavefront.json 2> gpurun_out/r05_soak2/err1.log; echo "mode 1 rc=$?"; tail -c 300 gpurun_out/r05_soak2/soak_one_wavefront.json; echo
tail -n 1 gpurun_out/r05_soak2/err0.log gpurun_out/r05_soak2/err1.log
