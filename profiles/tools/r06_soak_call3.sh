#!/bin/bash
# Round 6, third soak, on the no-unroll + runs-first build (fresh seeds): the three kernel selections and the pair form pinned on three wavefronts
mkdir -p gpurun_out/r06_soak3
timeout 600 python profiles/tools/soak.py 510000 12800 3600 2400 1 > gpurun_out/r06_soak3/soak_one_wavefront.json 2> gpurun_out/r06_soak3/err1.log; echo "mode 1 rc=$?"; tail -c 320 gpurun_out/r06_soak3/soak_one_wavefront.json; echo
timeout 600 python profiles/tools/soak.py 520000 12800 3600 2400 0 > gpurun_out/r06_soak3/soak_auto.json 2> gpurun_out/r06_soak3/err0.log; echo "mode 0 rc=$?"; tail -c 320 gpurun_out/r06_soak3/soak_auto.json; echo
timeout 600 python profiles/tools/soak.py 530000 6400 7200 1600 2 > gpurun_out/r06_soak3/soak_two_wavefronts.json 2> gpurun_out/r06_soak3/err2.log; echo "mode 2 rc=$?"; tail -c 320 gpurun_out/r06_soak3/soak_two_wavefronts.json; echo
BROTLIG_SOAK_GRID=3 timeout 600 python profiles/tools/soak.py 540000 9600 3600 1600 1 > gpurun_out/r06_soak3/soak_pair3.json 2> gpurun_out/r06_soak3/err3.log; echo "pair3 rc=$?"; tail -c 320 gpurun_out/r06_soak3/soak_pair3.json; echo
# the benchmark line once more, now that profiles/r06_final_hbm_traffic.json carries this kernel's disassembly hash (roofline.traffic attached)
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r06_soak3/bench.json 2> gpurun_out/r06_soak3/bench.err; cut -c1-300 gpurun_out/r06_soak3/bench.json
timeout 600 python bench.py --two-in-flight --no-cpu-baseline --no-alt-parse 2>> gpurun_out/r06_soak3/bench.err | tee gpurun_out/r06_soak3/bench_two_in_flight.json | cut -c1-200
grep -o '"two_in_flight": {[^}]*}' gpurun_out/r06_soak3/bench_two_in_flight.json
