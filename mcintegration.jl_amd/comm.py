"""Communicators for the per-iteration sum of the packed statistics buffer (the reference's
MPIreduceConfig! + MPIbcastConfig!, src/configuration.jl:264-321, src/main.jl:177-188)."""
import numpy as np


class LocalComm:
    """single process (mpi_nprocs() == 1)"""
    rank, size = 0, 1

    def all_reduce(self, engine):
        return None

    def sum_host(self, engine, v):
        """a few host doubles summed over the ranks (the lineage sums of a run of carried chains, statistics.Result)"""
        return v


class RcclComm:
    """RCCL inside the library: one ncclAllReduce(sum, f64) on the engine's stream, device to device.
    The communicator belongs to the mci_ctx of ONE device; an engine on another device would skip the reduction
    silently (mci_iteration_reduce is a no-op on a context without a communicator), so all_reduce checks it."""

    def __init__(self, rank, size, unique_id, device):
        import ctypes as C
        from ._lib import check, lib
        from .engine import context
        self.rank, self.size, self.device = rank, size, device
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

    def library_ranks(self):
        """(rank, nranks) as the library's communicator reports them (mci_comm_rank): what RCCL was initialised with"""
        import ctypes as C
        from ._lib import lib
        from .engine import context
        r, n = C.c_int32(), C.c_int32()
        lib().mci_comm_rank(context(self.device), C.byref(r), C.byref(n))
        return r.value, n.value

    def all_reduce(self, engine):
        if getattr(engine, "device", self.device) != self.device:
            raise RuntimeError("RcclComm was created for device %s but the engine runs on device %s: pass the same `device` to "
                               "integrate() (the all-reduce would be skipped silently)" % (self.device, engine.device))
        engine.reduce()

    def sum_host(self, engine, v):
        return engine.comm_sum(v)


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
        # (what is summed: the packed buffer and, behind it, the 64 :mcmc holding-time counts -- every rank sizes its next chains from the
        # same numbers; engines without them, e.g. the oracle-backed test engine, reduce the packed buffer alone)
        ext_size = hasattr(engine, "external_reduce_done")
        if self.tensor_device == "cpu":
            t = torch.from_numpy(np.ascontiguousarray(engine.get_packed(reduce_size=True) if ext_size else engine.get_packed()))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            engine.set_packed(t.numpy())
            if ext_size:
                engine.external_reduce_done()
            return
        ptr = engine.packed_device_ptr()
        if not ptr:
            raise RuntimeError("the engine has no device buffer to reduce")
        if self._view[0] != ptr:
            t = torch.as_tensor(_DeviceBuffer(ptr, engine.reduce_size if ext_size else engine.packed_size), device=self.tensor_device)
            self._view = (ptr, t, torch.cuda.ExternalStream(engine.stream(), device=self.tensor_device))
        _, t, ext = self._view
        with torch.cuda.stream(ext):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        if ext_size:
            engine.external_reduce_done()

    def sum_host(self, engine, v):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        if self.tensor_device != "cpu":
            t = t.to(self.tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()
