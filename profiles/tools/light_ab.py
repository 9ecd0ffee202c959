#!/usr/bin/env python3
"""The light-page kernel beside the main one (csrc/brotlig_light.hip) against the main kernel alone, in ONE process on one box: every
workload's streams are generated and encoded once, then decoded with the light kernel off (BrotligDebugSetLightKernel(1)), with the
product rule (0) and -- `--buckets` -- with other splits of the schedule, timed passes interleaved over the settings; every setting's
output is compared with the source bytes.  Needs BROTLIG_ENABLE_DEBUG_KNOBS=1 in the environment.

  BROTLIG_ENABLE_DEBUG_KNOBS=1 python profiles/tools/light_ab.py [--workloads mixed runs ...] [--buckets 24 32] [--reps 3] [--steps 5] [--out f.json]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="+", default=["mixed"])
    ap.add_argument("--buckets", nargs="*", type=int, default=[])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    import bench
    from brotli_g_sdk_amd import api
    settings = [("main_only", 1, 0), ("rule", 0, 0)] + [("bucket%d" % b if b < 256 else "probe%d_%d" % (b >> 8, b & 255), 0, b) for b in a.buckets]
    results = {}
    for spec in a.workloads:
        w, *shape = spec.split(":")
        nstreams = int(shape[0]) if shape else (256 if w == "bc3" else 16)
        npages = int(shape[1]) if len(shape) > 1 else (256 if w == "bc3" else 4096)
        streams, expected = bench.build_streams(w, list(range(nstreams)), npages, 8 if w == "bc3" else min(256, npages))
        out_sizes = [len(e) for e in expected] if w == "bc3" else None
        dec = api.BatchDecoder(streams, out_sizes=out_sizes)
        per = {n: [] for n, _, _ in settings}
        step = {n: [] for n, _, _ in settings}
        exact = {}
        for rep in range(a.reps):
            for n, mode, bucket in settings:
                api.DebugSetLightKernel(mode, bucket)
                if rep == 0:
                    dec.poison_output(); dec.decode(check=True); torch.cuda.synchronize()
                    ok = True
                    for k in range(len(streams)):
                        exp = torch.from_numpy(expected[k]).to(dec.device)
                        got = dec.d_out[dec.out_offs[k]:dec.out_offs[k] + dec.sizes[k]].view(-1, exp.numel())
                        ok = ok and bool((got == exp.unsqueeze(0)).all())
                    exact[n] = ok
                total, kern = dec.timed(2, a.steps)
                per[n].append(kern if w != "bc3" else total / a.steps)
                step[n].append(total / a.steps)
        api.DebugSetLightKernel(0, 0)
        U = dec.decompressed_bytes
        base = min(per[settings[0][0]])
        for n, _, _ in settings:
            best = min(per[n])
            results.setdefault(spec, {})[n] = {"ms": [round(x, 4) for x in per[n]], "best_ms": round(best, 4), "GBps": round(U / best / 1e6, 1),
                                              "vs_main_only_pct": round((base / best - 1) * 100, 2), "bit_exact": exact[n], "step_ms": round(min(step[n]), 4),
                                              "step_GBps": round(U / min(step[n]) / 1e6, 1)}
            print(f"{spec:12s} {n:12s} page kernels {best:8.4f} ms {U / best / 1e6:7.1f} GB/s {100 * (base / best - 1):+6.2f} %  step {min(step[n]):8.4f} ms ({U / min(step[n]) / 1e6:7.1f} GB/s) exact {exact[n]} runs {[round(x, 3) for x in per[n]]}", flush=True)
        del dec
        torch.cuda.empty_cache()
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(results, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
