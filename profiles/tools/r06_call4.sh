#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c4; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $out/pytest.log | tail -3
timeout 600 python profiles/tools/sched_times.py mixed runs:1 mixed:16:64 2>$out/err.log > $out/sched_times.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r06c4/sched_times.jsonl"):
    d = json.loads(l); print(d["workload"], d["tickets"], "total_us", d["first_came_to_last_done_us"], "ticket0", d["ticket0"], "last", d["last_ticket"])
PY
timeout 600 python profiles/tools/ab_run.py --workloads mixed runs:1 mixed:16:64 mixed:16:8 bc3 text --reps 3 --steps 10 --out $out/ab.json 2>>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
