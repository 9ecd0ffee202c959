#!/bin/bash
# Round 6: why did ab_run.py call a wrong build "exact"?  Two libraries, one workload, with the mismatching bytes counted.
export TMPDIR=/tmp
out=gpurun_out/r06c49; mkdir -p $out
python - <<'PY' 2>&1 | tail -20
import os, sys, glob
sys.path.insert(0, os.getcwd())
import torch, bench
from brotli_g_sdk_amd import api
streams, expected = bench.build_streams("runs", list(range(4)), 1024, 256)
for order in (["base", "short40"], ["short40", "base"]):
    for n in order:
        so = os.path.join(os.getcwd(), "build/abv2/lib_%s.so" % n)
        os.environ["BROTLIG_HIP_SO"] = so
        api._lib = None
        dec = api.BatchDecoder(streams)
        dec.poison_output(); dec.decode(check=True); torch.cuda.synchronize()
        bad = 0
        for k in range(len(streams)):
            exp = torch.from_numpy(expected[k]).to(dec.device)
            got = dec.d_out[dec.out_offs[k]:dec.out_offs[k] + dec.sizes[k]].view(-1, exp.numel())
            bad += int((got != exp.unsqueeze(0)).sum())
        total, kern = dec.timed(1, 3)
        print(order, n, "lib handle", api.lib()._cdll._name if hasattr(api.lib(), "_cdll") else "?", "mismatching bytes", bad, "kernel ms %.4f" % kern, flush=True)
        del dec
PY
