import os; os.environ.setdefault("BROTLIG_ENABLE_DEBUG_KNOBS", "1")
import sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from brotli_g_sdk_amd import api, datagen as D, encoder as E
res = {}
base = D.mixed(256 * 65536, 1); s256 = E.encode(base)
for pages in (512, 1024, 1536, 1792, 2048, 2304, 3072, 4096):
    s = D.tile_stream(s256, pages // 256)
    dec = api.BatchDecoder([s]); dec.decode()
    ks = [dec.timed(3, 20)[1] for _ in range(3)]
    res[pages] = round(min(ks), 4)
print(json.dumps(res))
