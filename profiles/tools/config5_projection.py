#!/usr/bin/env python3
"""BASELINE.json config 5 on ONE device: 32 GiB of 64 KiB pages (128 streams x 4096 pages, 'mixed' class) cut into 8 shards
by BrotligShardPlan, every shard decoded through BrotligDecodeBatchMultiDevice -- one after the other on the one visible
GPU -- and checked byte for byte.  The figure it writes is a PROJECTION, labelled as such: U_total / max_g t_g is what
eight devices would deliver if each ran its shard as fast as this one did (the shards exchange nothing: SURVEY.md 8(e) row 4,
7.0(iii); the reference's fan-out: src/BrotligDecoder.cpp:356-375, BrotliGCompute.hlsl:1757-1881).  It is not a scaling
measurement; the measured curve comes from `bench.py --gpus N` on a multi-GPU node.

  python profiles/tools/config5_projection.py [--shards 8] [--streams-per-shard 16] [--pages 4096] [--out profiles/r04_config5_projection.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(shards=8, streams_per_shard=16, pages=4096, distinct=256, steps=3, warmup=1, log=None):
    import numpy as np
    import torch
    import bench
    from brotli_g_sdk_amd import api
    n = shards * streams_per_shard
    # the compressed size of stream s is known before it is built only after encoding it once: encode every stream's distinct
    # pages first (what a host has on disk), then plan on those sizes (a tiled stream is `rep` times its distinct body)
    from brotli_g_sdk_amd import datagen as D
    small, source, t_enc = {}, {}, time.perf_counter()
    sizes = []
    rep = pages // distinct
    for s in range(n):
        st, ex = bench.build_streams("mixed", [s], distinct, distinct)
        small[s], source[s] = st[0], ex[0]
        sizes.append(8 + 4 * pages + (len(st[0]) - 8 - 4 * distinct) * rep)
    t_enc = time.perf_counter() - t_enc
    first = api.ShardPlan(sizes, shards)
    assert first[0] == 0 and first[-1] == n and all(b > a for a, b in zip(first, first[1:]))
    per_shard, total_u, total_c, ok_all = [], 0, 0, True
    for g in range(shards):
        idx = list(range(first[g], first[g + 1]))
        streams = [D.tile_stream(small[s], rep) if rep > 1 else small[s] for s in idx]
        assert [len(x) for x in streams] == [sizes[s] for s in idx]
        dec = api.BatchDecoder(streams, device="cuda:0")
        mk, mw, per = api.DecodeBatchMultiDevice([dec], warmup=warmup, steps=steps)
        assert per[0][0] == 0, per
        dec.poison_output()
        _, _, per2 = api.DecodeBatchMultiDevice([dec], warmup=0, steps=1)
        assert per2[0][0] == 0, per2
        torch.cuda.synchronize()
        ok = True
        for k, s in enumerate(idx):
            exp = torch.from_numpy(source[s]).to("cuda:0")
            got = dec.d_out[dec.out_offs[k]:dec.out_offs[k] + dec.sizes[k]].view(-1, exp.numel())   # every tiled repeat
            ok = ok and got.shape[0] == rep and bool((got == exp.unsqueeze(0)).all())
        ok_all = ok_all and ok
        per_shard.append({"shard": g, "streams": [idx[0], idx[-1]], "decompressed_bytes": int(dec.decompressed_bytes),
                          "compressed_bytes": int(dec.compressed_bytes), "kernel_ms": round(mk, 4), "wall_ms_per_step": round(mw / steps, 4),
                          "bit_exact": ok})
        total_u += int(dec.decompressed_bytes); total_c += int(dec.compressed_bytes)
        if log:
            log(f"shard {g}: streams {idx[0]}..{idx[-1]}  kernel {mk:.3f} ms  exact {ok}")
        del dec, streams
        torch.cuda.empty_cache()
    t_max = max(p["kernel_ms"] for p in per_shard)
    w_max = max(p["wall_ms_per_step"] for p in per_shard)
    return {
        "label": "projection, one device: each of the G shards decoded in turn on the one visible MI355X; NOT a multi-GPU measurement",
        "config": f"BASELINE.json configs[4]: {n} streams x {pages} pages x 64 KiB = {total_u / 2**30:.1f} GiB, 'mixed' synthetic, {shards} shards by BrotligShardPlan (compressed bytes)",
        "entry": "BrotligDecodeBatchMultiDevice (one shard per call, device 0)",
        "shards": per_shard,
        "total_decompressed_bytes": total_u, "total_compressed_bytes": total_c,
        "max_kernel_ms": t_max, "max_wall_ms_per_step": w_max,
        "projected_GBps_kernel": round(total_u / (t_max * 1e-3) / 1e9, 2),
        "projected_GBps_wall": round(total_u / (w_max * 1e-3) / 1e9, 2),
        "sum_kernel_ms_one_device": round(sum(p["kernel_ms"] for p in per_shard), 3),
        "one_device_GBps_kernel": round(total_u / (sum(p["kernel_ms"] for p in per_shard) * 1e-3) / 1e9, 2),
        "shard_imbalance_compressed": round(max(p["compressed_bytes"] for p in per_shard) * shards / total_c, 4),
        "bit_exact": ok_all, "encode_s": round(t_enc, 1), "kernel_source_sha16": bench.kernel_source_hash(),
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--streams-per-shard", type=int, default=16)
    ap.add_argument("--pages", type=int, default=4096)
    ap.add_argument("--distinct", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = run(a.shards, a.streams_per_shard, a.pages, a.distinct, a.steps, log=lambda m: print(m, file=sys.stderr, flush=True))
    text = json.dumps(r, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text + "\n")
    print(text)
    if not r["bit_exact"]:
        raise SystemExit("config 5 projection: output not bit-exact")
