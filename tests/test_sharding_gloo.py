"""N>1 path on CPU: world_size-2 gloo run of the stream sharding + result gather that
bench.py / a multi-GPU caller use (no data-path collective, one final
all_gather of per-stream results)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rustpotter_amd import sharding


def test_shard_bounds_cover_all_streams():
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = sharding.shard_bounds(total, world, r)
                assert lo == prev and hi >= lo
                prev = hi
                sizes.append(hi - lo)
            assert prev == total and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # weak scaling (bench.py): S streams per rank, global ids rank*S..
        S = 5
        first = sharding.weak_first_stream(S, rank)
        local = torch.arange(first, first + S, dtype=torch.int32) * 3 + 1  # stands in for n_det[stream]
        allv = sharding.gather_per_stream(local, world)
        assert torch.equal(allv, torch.arange(0, world * S, dtype=torch.int32) * 3 + 1)
        # strong scaling of a fixed stream set with ragged shards
        lo, hi = sharding.shard_bounds(total, world, rank)
        loc2 = torch.stack([torch.arange(lo, hi, dtype=torch.float32), torch.arange(lo, hi, dtype=torch.float32) * 0.5], dim=1)
        all2 = sharding.gather_ragged(loc2, world)
        assert all2.shape == (total, 2) and torch.equal(all2[:, 0], torch.arange(total, dtype=torch.float32))
        # the per-stream result block of SURVEY 8e (T + 2 floats per stream) through the same gather, weak and strong
        T, n_win = 3, 4
        sc = (torch.arange(first, first + S, dtype=torch.float32)[:, None, None] + torch.arange(n_win, dtype=torch.float32)[None, :, None] * 0.1 +
              torch.arange(T, dtype=torch.float32)[None, None, :] * 0.01)
        blk = sharding.stream_summary(sc, sc.amax(dim=2), local)
        assert blk.shape == (S, T + 2) and blk.dtype == torch.float32
        allb = sharding.gather_per_stream(blk, world)
        assert allb.shape == (world * S, T + 2)
        gs = torch.arange(world * S, dtype=torch.float32)
        assert torch.allclose(allb[:, 0], gs + 0.3) and torch.allclose(allb[:, T - 1], gs + 0.3 + 0.01 * (T - 1))   # best window of template 0 / T-1
        assert torch.allclose(allb[:, T], gs + 0.3 + 0.01 * (T - 1)) and torch.equal(allb[:, T + 1], (gs * 3 + 1))     # best aggregate, detections
        # timing reduction used by bench.py: MAX over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        # the self-proving record of a multi-rank bench line: world size as the group reports it, one entry per rank in rank
        # order; ranks that share a device are reported by a gloo dry run and REFUSED for an RCCL ("nccl") run
        rec = sharding.rank_identities({"rank": rank, "local_rank": rank, "device_index": rank, "uuid": "GPU-%04d" % rank}, "gloo")
        assert rec["rccl_world_size"] == world and rec["distinct_devices"] == world
        assert [e["rank"] for e in rec["rank_devices"]] == list(range(world)) and rec["rank_devices"][rank]["uuid"] == "GPU-%04d" % rank
        shared = sharding.rank_identities({"rank": rank, "uuid": "GPU-0000"}, "gloo")
        assert shared["distinct_devices"] == 1
        with pytest.raises(RuntimeError, match="share a device"):
            sharding.rank_identities({"rank": rank, "uuid": "GPU-0000"}, "nccl")   # backend NAME decides: the uuids are what RCCL ranks must differ in
        sharding.rank_identities({"rank": rank, "uuid": "GPU-%04d" % rank}, "nccl")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 11), nprocs=2, join=True)
