#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c12; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $out/pytest.log | tail -3
timeout 600 python profiles/tools/r05_many_streams.py 2>>$out/err.log | tee $out/many_streams.jsonl
timeout 300 python profiles/tools/streamer_bench.py > $out/streamer_bench.json 2>>$out/err.log; cat $out/streamer_bench.json
tail -3 $out/err.log
