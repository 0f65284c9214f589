#!/usr/bin/env python3
"""Can this torch/RCCL build capture collectives into a hipGraph?  (1-rank nccl group on one GPU: the
capture path through ProcessGroupNCCL is the one the N > 1 decode graphs use.)"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(16 * 7168, device="cuda", dtype=torch.bfloat16)
g2 = torch.ones(16, 1024, device="cuda", dtype=torch.bfloat16)
dist.all_reduce(t)  # communicator init outside capture
outs = [torch.empty_like(g2)]
dist.all_gather(outs, g2)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t.mul_(2)
    dist.all_reduce(t)
    dist.all_gather(outs, g2)
    t.add_(1)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("captured + replayed collectives OK:", float(t[0]), torch.__version__, torch.cuda.nccl.version())
dist.destroy_process_group()
