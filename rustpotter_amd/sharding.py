"""Stream sharding across the GPUs of one node.

Streams share nothing but the read-only templates, so rank g of G owns a contiguous
block of streams and the only exchange is one all_gather of the per-stream detection
summary at the end of a pass (RCCL over xGMI on GPUs; the same code runs on gloo/CPU
tensors in the tests)."""


def shard_bounds(total_streams, world_size, rank):
    """Contiguous [lo, hi) block of rank `rank`; blocks differ by at most one stream."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(total_streams, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def weak_first_stream(streams_per_gpu, rank):
    """Weak scaling (bench.py): every rank gets `streams_per_gpu` streams with global ids
    [rank*S, (rank+1)*S)."""
    return rank * streams_per_gpu


def stream_summary(scores, agg, n_det):
    """The per-stream result block a shard hands to the gather: T + 2 floats per stream --
    the best score of every template over the stream's windows, the best aggregate score, the number of detections.
    scores [S][n_win][T], agg [S][n_win], n_det [S] (torch tensors on one device) -> float32 [S][T + 2].
    At BASELINE config C4 (8 192 streams x 64 templates per GPU) that is 8 192 x 66 x 4 B = 2.16 MB per GPU."""
    import torch
    S, _, T = scores.shape
    out = torch.empty((S, T + 2), dtype=torch.float32, device=scores.device)
    out[:, :T] = torch.amax(scores, dim=1)
    out[:, T] = torch.amax(agg, dim=1)
    out[:, T + 1] = n_det.to(torch.float32)
    return out


def gather_per_stream(local, world_size, group=None, force_collective=False):
    """all_gather of an equally sized per-stream tensor -> tensor [world_size * S_local, ...]
    ordered by global stream id (rank-major, the order shard_bounds / weak_first_stream use).
    force_collective: a one-rank job still goes through the collective (the RCCL exercise on a one-GPU box, tests/test_gpu_rccl.py)."""
    import torch
    import torch.distributed as dist
    if world_size == 1 and not force_collective:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":  # CPU-backend dry runs: stage through host memory
        return gather_per_stream(local.cpu(), world_size, group).to(local.device)
    parts = [torch.empty_like(local) for _ in range(world_size)]
    dist.all_gather(parts, local, group=group)
    return torch.cat(parts, dim=0)


def gather_ragged(local, world_size, group=None, force_collective=False):
    """all_gather for shards whose sizes differ by at most one row (strong scaling of a fixed
    stream set): pads to the largest shard, gathers, trims.  force_collective: as gather_per_stream."""
    import torch
    import torch.distributed as dist
    if world_size == 1 and not force_collective:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":  # CPU-backend dry runs: stage through host memory
        return gather_ragged(local.cpu(), world_size, group).to(local.device)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world_size)]
    dist.all_gather(sizes, n, group=group)
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: int(s.item())] for p, s in zip(parts, sizes)], dim=0)


def rank_identities(mine, backend, group=None):
    """Self-proving record of a multi-rank run (bench.py puts it on every --gpus N line): every rank contributes `mine`
    (rank, local_rank, device index, device uuid, ...), all ranks get the list ordered by rank, the world size the process
    group itself reports, and the number of distinct device uuids.  With backend "nccl" (RCCL: one rank per GPU) the uuids
    must be `world_size` distinct non-empty devices -- ranks sharing a GPU would make a scaling number meaningless, so that
    is an error there, while a gloo dry run (ranks sharing devices on a one-GPU box) only reports it."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    everyone = [None] * world
    dist.all_gather_object(everyone, dict(mine), group=group)
    everyone.sort(key=lambda e: e["rank"])
    uuids = [str(e.get("uuid", "")) for e in everyone]
    distinct = len(set(uuids))
    if backend == "nccl" and (distinct != world or not all(uuids)):
        raise RuntimeError("RCCL run whose ranks share a device (or report no uuid): %r" % (uuids,))
    return {"rccl_world_size": world, "rank_devices": everyone, "distinct_devices": distinct}
