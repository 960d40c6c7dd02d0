"""Process-exit stress: engine + library RCCL communicator (single rank) + torch's own process group, then a normal interpreter exit.
MCI_NO_ATEXIT=1 skips the orderly release (mcintegration_jl_amd.shutdown at atexit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import mcintegration_jl_amd as mci
from mcintegration_jl_amd.comm import RcclComm
if os.environ.get("MCI_NO_ATEXIT"):
    import atexit
    atexit.unregister(mci.shutdown)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29641")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
r = mci.integrate(mci.catalog.x2y2(), var=mci.Continuous(0.0, 1.0), dof=[[2]], solver="vegas", neval=1e5, seed=4, comm=comm)
r2 = mci.integrate(mci.catalog.sphere2(), var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], solver="mcmc", neval=1e5, seed=4)
dist.destroy_process_group()
print("done", r.mean[0])
