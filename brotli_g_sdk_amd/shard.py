"""Multi-GPU fan-out of the decode path.  Pages (and therefore streams) are independent
(no cross-page references, per-page prefix codes), so the path shards with no
data-path collective: rank r of W decodes a contiguous slice of the stream list on its own GPU.
torch.distributed (RCCL on ROCm, gloo in the CPU tests) is used only for the barrier and the
timing/byte-count reductions of the benchmark."""


def shard_plan(in_sizes, num_shards):
    """Pure-Python twin of BrotligShardPlan (include/brotlig_amd.h, csrc/brotlig_shard_plan.h): the num_shards + 1 run
    boundaries.  Host arithmetic, so that a rank computes its shard without building or loading any native library (a
    gloo / CPU-only rank has no hipcc); tests/test_abi.py checks it against the C export of both libraries."""
    import bisect
    sizes = [int(x) for x in in_sizes]
    n, G = len(sizes), int(num_shards)
    if G <= 0:
        raise ValueError("num_shards must be positive")
    pre = [0]
    for x in sizes:
        pre.append(pre[-1] + x)

    def run_end(i, cap):            # end of the longest run from i that weighs at most cap (at least one stream)
        return max(bisect.bisect_right(pre, pre[i] + cap, i + 1) - 1, i + 1)

    def runs_needed(cap):
        runs, i = 0, 0
        while i < n:
            i = run_end(i, cap)
            runs += 1
        return runs

    lo, hi = max(sizes, default=0), pre[n]
    while lo < hi:                  # the bottleneck: smallest cap that packs into at most G runs
        mid = lo + (hi - lo) // 2
        if runs_needed(mid) <= G:
            hi = mid
        else:
            lo = mid + 1
    cap = lo
    need = [0] * (n + 1)            # runs needed for streams i.. under the cap
    for i in range(n - 1, -1, -1):
        need[i] = 1 + need[run_end(i, cap)]
    first, i = [], 0
    for g in range(G):
        first.append(i)
        after = G - 1 - g
        if i >= n:
            continue
        if after == 0:
            i = n
            continue
        e_max = min(run_end(i, cap), n - min(after, n - i - 1))
        e_min = i + 1
        while e_min < e_max and need[e_min] > after:
            e_min += 1
        ideal = (pre[n] - pre[i] + (G - g) - 1) // (G - g)      # an even share of what is left
        best = e_min
        for e in range(e_min, e_max + 1):
            if abs(pre[e] - pre[i] - ideal) <= abs(pre[best] - pre[i] - ideal):
                best = e
        i = best
    first.append(n)
    return first


def stream_indices(n_streams, world, rank, in_sizes=None):
    """Contiguous slice of range(n_streams) owned by `rank`: the run BrotligShardPlan (include/brotlig_amd.h) gives it --
    runs balanced by COMPRESSED bytes (`in_sizes`, one per stream; SURVEY.md 8(e): "balance by compressed bytes, not
    page count"), streams never split.  Without sizes every stream weighs the same."""
    sizes = [1] * n_streams if in_sizes is None else [int(x) for x in in_sizes]
    assert len(sizes) == n_streams
    first = shard_plan(sizes, world)
    return list(range(first[rank], first[rank + 1]))


def _reduce(value, op_name):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op_name))
    return float(t.item())


def max_over_ranks(value):
    return _reduce(value, "MAX")


def min_over_ranks(value):
    return _reduce(value, "MIN")


def sum_over_ranks(value):
    return int(_reduce(value, "SUM"))


# ---- optional exchange steps around the decode ----------------------------------
# Decoding needs no collective.  These two helpers cover the cases where the data does not start or
# end sharded: compressed streams that arrive on one rank, and consumers that want the whole output
# on every GPU.  Both are outside the timed region of bench.py unless --gather is given, and their
# cost is reported separately (an all-gather of 32 GiB over xGMI costs about as much as the decode).

def _device_for_backend():
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def scatter_streams(streams, src=0):
    """`streams` (list of uint8 numpy arrays) is only meaningful on rank `src`; every rank returns the
    slice of it that stream_indices() assigns to it.  Sizes travel first (broadcast), then one
    point-to-point transfer per destination rank, grouped (ncclSend/ncclRecv under RCCL)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(streams)
    world, rank, dev = dist.get_world_size(), dist.get_rank(), _device_for_backend()
    n = torch.tensor([len(streams) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    sizes = torch.zeros(int(n.item()), dtype=torch.int64, device=dev)
    if rank == src:
        sizes.copy_(torch.tensor([len(s) for s in streams], dtype=torch.int64))
    dist.broadcast(sizes, src)
    sizes = sizes.cpu().tolist()
    mine = stream_indices(len(sizes), world, rank, sizes)
    if rank == src:
        ops, keep = [], []
        for r in range(world):
            idx = stream_indices(len(sizes), world, r, sizes)
            if r == src or not idx:
                continue
            buf = torch.from_numpy(np.concatenate([np.asarray(streams[i], dtype=np.uint8) for i in idx])).to(dev)
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, r))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return [np.asarray(streams[i], dtype=np.uint8) for i in mine]
    total = sum(sizes[i] for i in mine)
    if total == 0:
        return []
    buf = torch.empty(total, dtype=torch.uint8, device=dev)
    for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, src)]):
        w.wait()
    host, out, pos = buf.cpu().numpy(), [], 0
    for i in mine:
        out.append(host[pos:pos + sizes[i]].copy())
        pos += sizes[i]
    return out


def gather_outputs(local):
    """All-gather of each rank's decoded bytes (a 1-D uint8 tensor on the backend's device; lengths may
    differ).  Returns (gathered [world, padded_len] tensor, list of true lengths)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.reshape(1, -1), [local.numel()]
    world = dist.get_world_size()
    lens = torch.zeros(world, dtype=torch.int64, device=local.device)
    lens[dist.get_rank()] = local.numel()
    dist.all_reduce(lens, op=dist.ReduceOp.SUM)
    padded = int(lens.max().item())
    send = local if local.numel() == padded else torch.cat([local, local.new_zeros(padded - local.numel())])
    out = torch.empty(world * padded, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, send.contiguous())
    return out.reshape(world, padded), lens.cpu().tolist()
