"""One process, one rank: torch.distributed on "nccl" (torch's RCCL) and the library's own communicator (dvn_comm_*, librccl
opened by the library) side by side -- do both initialise and all-gather?  Usage: python tools/gpu_comm_coexist.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29377", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.arange(12, dtype=torch.int64, device="cuda"); r = torch.empty(12, dtype=torch.int64, device="cuda")
dist.all_gather_into_tensor(r, t); torch.cuda.synchronize()
ctx = pkg.Context(0)
comm = pkg.Comm(ctx, pkg.Comm.unique_id(), 0, 1)
rows = comm.all_gather(np.arange(12, dtype=np.uint64))
dist.all_gather_into_tensor(r, t); torch.cuda.synchronize()
print("torch:", r.cpu().tolist() == list(range(12)), "library:", rows[0].tolist() == list(range(12)), flush=True)
comm.free(); dist.barrier(); dist.destroy_process_group()
print("COEXIST_OK", flush=True)
