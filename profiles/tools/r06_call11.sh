#!/bin/bash
# Round 6: the batches of many small assets and the small batches on the schedule kernel + job records.
export TMPDIR=/tmp
out=gpurun_out/r06c11; mkdir -p $out
timeout 600 python profiles/tools/r05_many_streams.py 2>>$out/err.log | tee $out/many_streams.jsonl
timeout 600 python profiles/tools/r05_many_textures.py 2>>$out/err.log | tee $out/many_textures.jsonl
timeout 600 python profiles/tools/latency.py auto 2>>$out/err.log | tee $out/latency.json | cut -c1-1500
tail -3 $out/err.log
