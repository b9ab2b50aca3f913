"""One-rank probe of the RCCL plumbing bench.py uses at N > 1 (process group with a bound device, a second
communicator, all-gather / all-reduce / barrier).  Peer-to-peer needs two GPUs and is not covered."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g2 = dist.new_group(ranks=[0])
x = torch.arange(6, dtype=torch.int64, device=dev)
out = torch.empty(6, dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(out, x, group=g2)
t = torch.tensor([3.5], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert out.tolist() == list(range(6)) and t.item() == 3.5
print("rccl probe ok:", torch.cuda.nccl.version())
dist.destroy_process_group()
