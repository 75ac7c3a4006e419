"""Probe: can two ranks share GPU 0 under RCCL on this box?  (multi-GPU paths cannot otherwise be exercised on a 1-GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
from mcptam_amd.dist import init_rccl_comm
try:
    comm = init_rccl_comm(rank, world, 0)
    t = torch.full((1000,), float(rank + 1), dtype=torch.float64, device="cuda")
    comm.allreduce(t.data_ptr(), t.numel())
    torch.cuda.synchronize()
    print("rank", rank, "native comm allreduce ->", t[0].item(), flush=True)
    comm.close()
except Exception as e:
    print("rank", rank, "native comm failed:", repr(e), flush=True)
dist.destroy_process_group()
