#!/usr/bin/env python3
"""A/B of decode-kernel builds on one box, in one process: every workload's streams are generated and encoded ONCE, then
each library of build/abv/ (profiles/tools/ab_variants.sh build ...) decodes them -- timed passes interleaved over the
variants (round-robin, `--reps` rounds) so that clock drift of the box hits all of them alike -- and every variant's output
is compared with the source bytes.

  python profiles/tools/ab_run.py [--workloads mixed text ...] [--reps 3] [--steps 5] [--out gpurun_out/abv/summary.json]
"""
import argparse, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="+", default=["mixed"])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dir", default=os.path.join(ROOT, "build", "abv"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--encoder-flags", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=256, help="distinct encoded pages per stream (4096 = no tiling at all)")
    a = ap.parse_args()
    import torch
    import bench
    from brotli_g_sdk_amd import api
    bench.ENCODER_FLAGS = a.encoder_flags
    libs = sorted(glob.glob(os.path.join(a.dir, "lib_*.so")), key=lambda p: (os.path.basename(p) != "lib_base.so", p))
    names = [os.path.basename(p)[4:-3] for p in libs]
    results = {}
    for spec in a.workloads:
        # "mixed", or "runs:1" (streams), or "mixed:16:64" (streams, pages per stream): round 6, for what a step costs beside its page kernel
        w, *shape = spec.split(":")
        nstreams = int(shape[0]) if shape else (256 if w == "bc3" else 16)
        npages = int(shape[1]) if len(shape) > 1 else (256 if w == "bc3" else 4096)
        streams, expected = bench.build_streams(w, list(range(nstreams)), npages, 8 if w == "bc3" else min(a.distinct, npages))
        w = spec
        out_sizes = [len(e) for e in expected] if w == "bc3" else None
        per = {n: [] for n in names}
        step = {n: [] for n in names}
        rest = {}
        exact = {}
        for rep in range(a.reps):
            for n, so in zip(names, libs):
                os.environ["BROTLIG_HIP_SO"] = so
                api._lib = None
                dec = api.BatchDecoder(streams, out_sizes=out_sizes)
                if rep == 0:
                    dec.poison_output(); dec.decode(check=True); torch.cuda.synchronize()
                    ok = True
                    for k in range(len(streams)):
                        exp = torch.from_numpy(expected[k]).to(dec.device)
                        got = dec.d_out[dec.out_offs[k]:dec.out_offs[k] + dec.sizes[k]].view(-1, exp.numel())
                        ok = ok and bool((got == exp.unsqueeze(0)).all())
                    exact[n] = ok
                total, kern = dec.timed(2, a.steps)
                per[n].append(kern if not w.startswith("bc3") else total / a.steps)
                step[n].append(total / a.steps)
                if w.startswith("bc3"):      # what the step takes beside the page kernel: prepare + de-conditioning
                    rest.setdefault(n, []).append(total / a.steps - kern)
                U = dec.decompressed_bytes
                del dec
                torch.cuda.empty_cache()
        base = min(per[names[0]])
        for n in names:
            best = min(per[n])
            results.setdefault(w, {})[n] = {"ms": [round(x, 4) for x in per[n]], "best_ms": round(best, 4), "GBps": round(U / best / 1e6, 1),
                                            "vs_base_pct": round((base / best - 1) * 100, 2), "bit_exact": exact[n],
                                            "step_ms": round(min(step[n]), 4), "step_minus_kernel_us": round((min(step[n]) - best) * 1e3, 1) if not w.startswith("bc3") else None,
                                            "step_GBps": round(U / min(step[n]) / 1e6, 1)}
            if n in rest:
                results[w][n]["step_minus_page_kernel_ms"] = round(min(rest[n]), 4)
                print(f"{w:10s} {n:28s} step - page kernel (prepare + de-conditioning): {min(rest[n]):.4f} ms", flush=True)
            print(f"{w:14s} {n:24s} best {best:8.4f} ms  {U / best / 1e6:7.1f} GB/s  {100 * (base / best - 1):+6.2f} %  step {min(step[n]):8.4f} ms ({U / min(step[n]) / 1e6:7.1f} GB/s)  exact {exact[n]}  runs {[round(x, 3) for x in per[n]]}", flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(results, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
