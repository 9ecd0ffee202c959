#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c7; mkdir -p $out; root=$(pwd)
( timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "streamer" ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 300 python profiles/tools/streamer_bench.py > $out/streamer_bench.json 2>>$out/err.log; cat $out/streamer_bench.json
for k in "mixed 16" "files 16" "text 16"; do timeout 300 python profiles/phase_profile.py $k 2>>$out/err.log; done > $out/phase_profile.jsonl
python - <<'PY'
import json
keys=["setup","tables","commands","ring","positions","literals","group_setup","level_tail","slide","pieces_and_far_loads","bitmaps","lv_short","lv_bytes","lv_long_and_far","lv_overlap","cmd_symbol","cmd_extra_bits"]
for l in open("gpurun_out/r06c7/phase_profile.jsonl"):
    d=json.loads(l)
    print(d["workload"], "rounds/page", round(d["rounds"]/d["pages"],1), "cyc/round", d["cycles_per_round"], "lv/round", d["levels_per_round"], "groups/round", round(d["groups"]/d["rounds"],2), "lit_steps/round", round(d["lit_steps"]/d["rounds"],2), "solo", round(d["solo_rounds"]/d["rounds"],3), "h/lvl", d["halves_per_level"], {k:d[k] for k in keys})
PY
tail -3 $out/err.log
