"""N > 1 path on CPU: two gloo ranks shard a list of streams the way bench.py shards work across
GPUs (independent streams per rank, no data-path collective), decode their shard with the oracle
standing in for the device, and the gathered digests must equal the single-process result."""
import hashlib
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from helpers import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from brotli_g_sdk_amd import datagen as D, encoder as E, shard
    from helpers import oracle_decode
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 6
    mine = shard.stream_indices(n_total, world, rank)
    digests = torch.zeros(n_total, 32, dtype=torch.uint8)
    nbytes = 0
    for k in mine:
        data = D.mixed(2 * 65536 + 17 * k, 200 + k)
        rc, out = oracle_decode(E.encode(data))
        assert rc == 0 and np.array_equal(out, data)
        digests[k] = torch.frombuffer(bytearray(hashlib.sha256(out.tobytes()).digest()), dtype=torch.uint8)
        nbytes += len(out)
    ms = shard.max_over_ranks(10.0 + rank)                 # bench.py's timing reduction
    total = shard.sum_over_ranks(nbytes)
    dist.all_reduce(digests, op=dist.ReduceOp.SUM)         # test-side gather only; disjoint rows
    if rank == 0:
        q.put((ms, total, digests.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_matches_single_process():
    from brotli_g_sdk_amd import datagen as D, shard
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ms, total, blob = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ms == 11.0
    expect = b"".join(hashlib.sha256(D.mixed(2 * 65536 + 17 * k, 200 + k).tobytes()).digest() for k in range(6))
    assert blob == expect
    assert total == sum(2 * 65536 + 17 * k for k in range(6))
    # shards are a disjoint cover
    cover = sorted(i for r in range(world) for i in shard.stream_indices(6, world, r))
    assert cover == list(range(6))


def test_shard_is_contiguous_and_balanced():
    from brotli_g_sdk_amd import shard
    for n in (1, 7, 16, 128):
        for world in (1, 2, 4, 8):
            parts = [shard.stream_indices(n, world, r) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from brotli_g_sdk_amd import datagen as D, encoder as E, shard
    from helpers import oracle_decode
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    datas = [D.mixed(65536 + 1000 * k, 300 + k) for k in range(5)]
    streams = [E.encode(d) for d in datas] if rank == 0 else []          # the compressed data starts on rank 0 only
    mine = shard.scatter_streams(streams, src=0)
    # (which rank gets which streams follows their compressed sizes: BrotligShardPlan; the gather below checks the cover)
    outs = []
    for s in mine:                                                       # the oracle stands in for the device here
        rc, out = oracle_decode(s)
        assert rc == 0
        outs.append(out)
    local = torch.from_numpy(np.concatenate(outs) if outs else np.zeros(0, np.uint8))
    gathered, lens = shard.gather_outputs(local)
    whole = np.concatenate([gathered[r, :lens[r]].numpy() for r in range(world)])
    assert np.array_equal(whole, np.concatenate(datas))                  # every rank ends up with every byte
    if rank == 0:
        q.put(hashlib.sha256(whole.tobytes()).hexdigest())
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_streams_and_gather_outputs():
    """The optional exchange steps: compressed streams scattered from one rank, decoded
    shards all-gathered; 2 gloo ranks, uneven shard sizes."""
    from brotli_g_sdk_amd import datagen as D
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    digest = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = hashlib.sha256(np.concatenate([D.mixed(65536 + 1000 * k, 300 + k) for k in range(5)]).tobytes()).hexdigest()
    assert digest == expect


# ---- the product path under more than one rank (GPU box) ------------------------------------------------------
import pytest  # noqa: E402


def _gpu_rank_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from brotli_g_sdk_amd import api, datagen as D, encoder as E, shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only; RCCL refuses two ranks on one device
    torch.cuda.set_device(0)
    n_total = 7
    mine = shard.stream_indices(n_total, world, rank)
    datas = {k: D.mixed(3 * 65536 + 1000 * k, 400 + k) for k in mine}
    dec = api.BatchDecoder([E.encode(datas[k]) for k in mine], device="cuda:0")      # the HIP path, not the oracle
    dec.poison_output()
    dec.decode()
    digests = torch.zeros(n_total, 32, dtype=torch.uint8)
    for i, k in enumerate(mine):
        out = dec.output(i)
        assert np.array_equal(out, datas[k])
        digests[k] = torch.frombuffer(bytearray(hashlib.sha256(out.tobytes()).digest()), dtype=torch.uint8)
    total = shard.sum_over_ranks(dec.decompressed_bytes)
    ranks = shard.sum_over_ranks(1)
    dist.all_reduce(digests, op=dist.ReduceOp.SUM)
    if rank == 0:
        q.put((total, ranks, digests.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_decode_their_shards_on_the_device():
    """Two processes, each decoding its shard.stream_indices slice through the C ABI on cuda:0 (the box has one
    GPU; the ranks share it, with a gloo control plane).  What the ranks produce together equals the source."""
    from brotli_g_sdk_amd import datagen as D
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    total, ranks, blob = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ranks == 2
    assert total == sum(3 * 65536 + 1000 * k for k in range(7))
    assert blob == b"".join(hashlib.sha256(D.mixed(3 * 65536 + 1000 * k, 400 + k).tobytes()).digest() for k in range(7))


@pytest.mark.gpu
def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks (torch.distributed.run on 127.0.0.1).
    With one device on the box the ranks are stacked on it (--stack-ranks: launch-path test, labelled as such);
    without that flag and with fewer devices than ranks it refuses instead of printing an n_gpus it did not use."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "2",
            "--pages-per-stream", "256", "--distinct", "64", "--no-cpu-baseline"]
    r = subprocess.run(base + ["--stack-ranks"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["bit_exact"] is True
    assert line["ranks"]["world_size"] == 2 and line["ranks"]["ranks_that_decoded"] == 2
    import torch
    if torch.cuda.device_count() < 2:
        assert line["ranks"]["stacked"] is True and line["n_gpus"] == torch.cuda.device_count()
        r = subprocess.run(base, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "only 1 HIP device" in (r.stderr + r.stdout)
