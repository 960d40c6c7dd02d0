"""Communicators for the per-iteration sum of the packed statistics buffer (the reference's
MPIreduceConfig! + MPIbcastConfig!, src/configuration.jl:264-321, src/main.jl:177-188)."""
import numpy as np


class LocalComm:
    """single process (mpi_nprocs() == 1)"""
    rank, size = 0, 1

    def all_reduce(self, engine):
        return None


class RcclComm:
    """RCCL inside the library: one ncclAllReduce(sum, f64) on the engine's stream, device to device."""

    def __init__(self, rank, size, unique_id, device):
        import ctypes as C
        from ._lib import check, lib
        from .engine import context
        self.rank, self.size = rank, size
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(lib().mci_comm_init(context(device), rank, size, buf))

    @staticmethod
    def unique_id():
        import ctypes as C
        from ._lib import check, lib
        buf = C.create_string_buffer(128)
        check(lib().mci_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def from_torch_distributed(cls, device):
        """bootstrap: rank 0 creates the id, torch.distributed ships it (any backend)."""
        import torch.distributed as dist
        rank, size = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(rank, size, box[0], device)

    def all_reduce(self, engine):
        engine.reduce()


class TorchDistComm:
    """External reducer through torch.distributed (gloo on CPU tensors, or nccl == RCCL on the GPU):
    packed buffer out of the engine, all_reduce(SUM), back in."""

    def __init__(self, group=None, tensor_device="cpu"):
        import torch.distributed as dist
        self.group, self.tensor_device = group, tensor_device
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce(self, engine):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(engine.get_packed()))
        if self.tensor_device != "cpu":
            t = t.to(self.tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        engine.set_packed(t.cpu().numpy())
