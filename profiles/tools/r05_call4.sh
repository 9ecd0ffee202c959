#!/bin/bash
# Round 5, fourth call: the GPU suite on the current default build, small-batch latencies with and without the producer-side dependency pass.
out=gpurun_out/r05c4; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $out/pytest.log | tail -3
for v in duo0 duo1; do BROTLIG_HIP_SO=$root/build/duo/lib_$v.so timeout 150 python profiles/tools/latency.py two_wavefronts 2>> $out/lat.err > $out/latency_$v.json; done
python - $out <<'PY'
import json, sys
out = sys.argv[1]
a, b = (json.loads(open(f"{out}/latency_{v}.json").read().strip().splitlines()[-1]) for v in ("duo0", "duo1"))
for k in a:
    if "kernel_ms" in a[k] and k in b and "asset" not in k:
        print(f"{k:36s} {a[k]['kernel_ms']:8.4f} -> {b[k]['kernel_ms']:8.4f} ms  {100 * (a[k]['kernel_ms'] / b[k]['kernel_ms'] - 1):+5.1f} %")
PY
