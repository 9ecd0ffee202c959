#!/bin/bash
# the batch that faulted, then the soak in both kernel selections on fresh seeds
mkdir -p gpurun_out/r05_soak
seeds=$(python -c "print(','.join(str(s) for s in range(260800,260880)))")
python profiles/tools/r05_bisect_dev.py run 1 $seeds 0 after 2>&1 | tail -n 1
python profiles/tools/r05_bisect_dev.py run 0 $seeds 0 after 2>&1 | tail -n 1
timeout 200 python profiles/tools/soak.py 260000 6400 1800 1200 1 > gpurun_out/r05_soak/soak_one_wavefront.json 2> gpurun_out/r05_soak/err1.log; echo "mode 1 rc=$?"; tail -c 320 gpurun_out/r05_soak/soak_one_wavefront.json; echo
timeout 200 python profiles/tools/soak.py 270000 6400 1800 1200 0 > gpurun_out/r05_soak/soak_auto.json 2> gpurun_out/r05_soak/err0.log; echo "mode 0 rc=$?"; tail -c 320 gpurun_out/r05_soak/soak_auto.json; echo
timeout 200 python profiles/tools/soak.py 280000 3200 3600 800 2 > gpurun_out/r05_soak/soak_two_wavefronts.json 2> gpurun_out/r05_soak/err2.log; echo "mode 2 rc=$?"; tail -c 320 gpurun_out/r05_soak/soak_two_wavefronts.json; echo
