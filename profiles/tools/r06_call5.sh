#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c6; mkdir -p $out
timeout 600 python profiles/tools/sched_times.py mixed runs:1 mixed:16:64 2>$out/err.log > $out/sched_times.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r06c6/sched_times.jsonl"):
    d = json.loads(l); print(d["workload"], d["tickets"], "total_us", d["first_came_to_last_done_us"], "published", d.get("published_us"), "ticket0", d["ticket0"], "last", d["last_ticket"], [(b["came_us"], b["open_us"], b["done_us"]) for b in d["by_ticket_block"]])
PY
timeout 600 python profiles/tools/ab_run.py --workloads mixed runs:1 mixed:16:64 bc3 --reps 3 --steps 10 --out $out/ab.json 2>>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
