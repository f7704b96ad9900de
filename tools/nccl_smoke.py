"""Developer tool: RCCL sanity on one GPU (world_size 1): the collectives ShardedPF uses with the "nccl" backend."""
import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
dev = torch.device("cuda", 0)
send = torch.arange(8, dtype=torch.float64, device=dev)
out = [torch.empty(8, dtype=torch.float64, device=dev)]
dist.all_gather(out, send)
assert torch.equal(out[0], send)
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
s = torch.zeros(3, dtype=torch.int64, device=dev); dist.all_reduce(s, op=dist.ReduceOp.SUM)
dist.barrier()
print("rccl world=1 ok:", dist.get_backend(), float(t.item()))
dist.destroy_process_group()
