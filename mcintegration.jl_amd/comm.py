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


class _DeviceBuffer:
    """a raw device pointer dressed for torch.as_tensor (zero copy)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchDistComm:
    """External reducer through torch.distributed: all_reduce(SUM) of the packed buffer.
    tensor_device="cpu" (gloo): packed buffer out of the engine, reduced on the host, back in.
    tensor_device="cuda:N" (nccl == RCCL): zero copy -- the engine's device buffer is wrapped as a tensor and the
    collective is issued with the library's stream current, so it is ordered between the sample pass and the
    refinement without any host synchronisation."""

    def __init__(self, group=None, tensor_device="cpu"):
        import torch.distributed as dist
        self.group, self.tensor_device = group, tensor_device
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self._view = (None, None, None)   # (ptr, tensor, external stream)

    def all_reduce(self, engine):
        import torch
        import torch.distributed as dist
        if self.tensor_device == "cpu":
            t = torch.from_numpy(np.ascontiguousarray(engine.get_packed()))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            engine.set_packed(t.numpy())
            return
        ptr = engine.packed_device_ptr()
        if not ptr:
            raise RuntimeError("the engine has no device buffer to reduce")
        if self._view[0] != ptr:
            t = torch.as_tensor(_DeviceBuffer(ptr, engine.packed_size), device=self.tensor_device)
            self._view = (ptr, t, torch.cuda.ExternalStream(engine.stream(), device=self.tensor_device))
        _, t, ext = self._view
        with torch.cuda.stream(ext):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
