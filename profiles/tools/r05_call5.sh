#!/bin/bash
# Is the 64-page outlier of the producer-side dependency pass real?  The same batch five times per build, builds alternating.
out=gpurun_out/r05c5; mkdir -p $out
root=$(pwd)
cat > /tmp/lat64.py <<'PY'
import os; os.environ.setdefault("BROTLIG_ENABLE_DEBUG_KNOBS", "1")
import sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from brotli_g_sdk_amd import api, datagen as D, encoder as E
api.DebugSetDecodeMode(2)
res = {}
for pages in (16, 32, 48, 64, 96, 128, 256):
    base = D.mixed(pages * 65536, 1)
    s = E.encode(base)
    dec = api.BatchDecoder([s])
    dec.decode()
    ks = []
    for _ in range(5):
        tot, k = dec.timed(3, 20); ks.append(round(k, 4))
    assert np.array_equal(dec.output(0), base)
    res[pages] = ks
print(json.dumps(res))
PY
for rep in 1 2; do for v in duo0 duo1; do echo -n "$v "; BROTLIG_HIP_SO=$root/build/duo/lib_$v.so timeout 120 python /tmp/lat64.py 2>> $out/err.log; done; done | tee $out/lat64.txt
