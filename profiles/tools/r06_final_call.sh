#!/bin/bash
# Round 6 collection: GPU suite, then the per-round profile set (profiles/tools/collect.sh r06_final).
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_final
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06_final/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_final/pytest.log | tail -2
bash profiles/tools/collect.sh r06_final > gpurun_out/r06_final/collect.log 2>&1; echo "collect rc=$?"
tail -1 gpurun_out/r06_final/bench.json | cut -c1-400
tail -5 gpurun_out/r06_final/bench.err
