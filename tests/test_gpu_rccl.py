"""RCCL on the one GPU a test box has (the only exchange of the path is the final per-stream gather).  Until a multi-GPU box
runs tools/run_scale.sh, this is what has executed: ncclCommInit through torch.distributed ("nccl" IS RCCL on ROCm) with world_size 1,
the per-stream result block (sharding.stream_summary: T + 2 floats per stream, reduced from DEVICE score arrays the library produced)
through all_gather on the NON-early-return path of sharding.gather_per_stream / gather_ragged, rank_identities, and bench.py's own
one-rank run with RP_BENCH_FORCE_PG=1 reporting `backend: nccl`.  Child processes: a process group is created before anything else
touches the GPU, and the parent test process never initialises RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
import numpy as np
import rustpotter_amd as ra
from rustpotter_amd import sharding
from oracle import rp_oracle as orc
SEED = 0x5EED000000000001
S, N, K, T, L = 64, 480 * 40, 5, 8, 40
ctx = ra.BatchContext(device=0, host_pointers=False)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
templates = orc.synth_templates(SEED, T, L, K)
tm = ra.Templates(ctx, templates)
pcm = torch.empty((S, N), dtype=torch.float32, device=dev)
ctx.synth_dev(SEED, 0, S, N, N, pcm.data_ptr())
nf = ra.mfcc_num_frames(N); n_win = nf - L + 1
scores = torch.empty((S, n_win, T), dtype=torch.float32, device=dev)
agg = torch.empty((S, n_win), dtype=torch.float32, device=dev)
det = torch.zeros((S, 4, 6), dtype=torch.int32, device=dev)
n_det = torch.zeros((S,), dtype=torch.int32, device=dev)
cfg = ra.DetectorConfig()
ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tm, cfg, det.data_ptr(), n_det.data_ptr(), 4, scores.data_ptr(), agg.data_ptr())
torch.cuda.synchronize()
block = sharding.stream_summary(scores, agg, n_det)
assert block.is_cuda and block.shape == (S, T + 2)
got = sharding.gather_per_stream(block, 1, force_collective=True)      # dist.all_gather on device tensors: RCCL
assert got.is_cuda and got.data_ptr() != block.data_ptr() and torch.equal(got, block)
got2 = sharding.gather_ragged(block[:S - 1], 1, force_collective=True)  # the strong-scaling form (size exchange + padded gather)
assert torch.equal(got2, block[:S - 1])
t = torch.tensor([2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                               # bench.py's max-over-ranks
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 2.5
p = torch.cuda.get_device_properties(dev)
ids = sharding.rank_identities({"rank": 0, "local_rank": 0, "device_index": 0, "uuid": str(getattr(p, "uuid", "")), "pid": os.getpid()}, "nccl")
assert ids["rccl_world_size"] == 1 and ids["distinct_devices"] == 1
# the block is what the oracle says: best score of template 0 of stream 0
mf = orc.mfcc_stream(orc.synth_pcm(SEED, 0, N), K)
ref = orc.score_stream(mf, templates)[0]
assert abs(float(block[0, 0].item()) - float(ref[:, 0].max())) <= 1e-5 * float(ref[:, 0].max())
dist.destroy_process_group()
print("rccl gather ok: %%d streams x %%d floats through all_gather" %% (S, T + 2))
"""


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)


def test_per_stream_block_through_rccl_world_size_one(tmp_path):
    script = tmp_path / "rccl_child.py"
    script.write_text(CHILD % {"root": ROOT})
    r = _run([sys.executable, str(script)])
    assert r.returncode == 0 and "rccl gather ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_one_rank_with_process_group():
    """bench.py --gpus 1 with RP_BENCH_FORCE_PG=1: the line reports backend nccl, the process group's world size and the gather it timed."""
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--streams", "2048", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
             env={"RP_BENCH_FORCE_PG": "1", "MASTER_PORT": "29543"})
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    c = j["config"]
    assert c["backend"] == "nccl" and c["rccl_world_size"] == 1 and c["world_size"] == 1
    g = c["gather_ms_per_step"]
    assert g["gathered_shape"] == [2048, 10] and g["mean"] > 0
    assert j["n_gpus"] == 1 and j["value"] > 0
