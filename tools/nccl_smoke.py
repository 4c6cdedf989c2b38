"""One-rank RCCL sanity check of the exchange bench.py does after a pass (world_size 1 so that it runs on a 1-GPU box):
init_process_group("nccl"), all_gather of a per-stream int32 tensor on the launch stream, barrier."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(65536, dtype=torch.int32, device=dev)
parts = [torch.empty_like(x)]
dist.all_gather(parts, x)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(parts[0], x)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("rccl ok")
