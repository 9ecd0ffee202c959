#!/bin/bash
# Round 6: kernel time line of the two page kernels (rocprofv3 --kernel-trace: start / end of every dispatch) on text and runs.
export TMPDIR=/tmp
out=gpurun_out/r06c32; mkdir -p $out
cd /tmp
for w in text runs; do
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace_$w -o f -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-alt-parse > $GRAFT_REPO_ROOT/$out/trace_$w.log 2>&1
  tail -2 $GRAFT_REPO_ROOT/$out/trace_$w.log | cut -c1-300
  f=$(ls $GRAFT_REPO_ROOT/$out/trace_$w/*/f_kernel_trace.csv 2>/dev/null | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-16:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print("%-60s q%-3s start %10.3f us  dur %10.3f us  grid %s" % (r["Kernel_Name"][:60], r.get("Queue_Id","?"), (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size","?"))))
PY
done
