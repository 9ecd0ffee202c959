"""Multi-GPU fan-out of the decode path.  Pages (and therefore streams) are independent
(no cross-page references, per-page prefix codes), so the path shards with no
data-path collective: rank r of W decodes a contiguous slice of the stream list on its own GPU.
torch.distributed (RCCL on ROCm, gloo in the CPU tests) is used only for the barrier and the
timing/byte-count reductions of the benchmark."""


def stream_indices(n_streams, world, rank):
    """Contiguous, balanced slice of range(n_streams) owned by `rank`."""
    base, rem = divmod(n_streams, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


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


def sum_over_ranks(value):
    return int(_reduce(value, "SUM"))
