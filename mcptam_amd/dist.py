"""All-reduce hooks for the sharded ChainBundle solve.

The C-ABI library is collective-agnostic: `mcp_ba_set_allreduce` installs a callback that sums
`count` doubles in place at a device pointer.  In production the callback is RCCL over xGMI
through torch.distributed (backend "nccl" IS RCCL on ROCm), one process per GPU.  A gloo
variant (host staging) exists so the sharded algorithm can be tested with several processes
sharing one GPU, and on pure host memory in the CPU test-suite.
"""
import ctypes

import numpy as np

hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice = 1, 2, 3


def _hip():
    # torch (if imported) has already loaded libamdhip64.so.7; the soname resolves to that copy
    for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            h = ctypes.CDLL(name)
            h.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            h.hipMemcpy.restype = ctypes.c_int
            h.hipDeviceSynchronize.restype = ctypes.c_int
            return h
        except OSError:
            continue
    raise RuntimeError("HIP runtime not found")


class RcclAllReduce:
    """fn(ptr, count, stream): SUM all-reduce over the default process group (backend nccl = RCCL)."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = device
        self.hip = _hip()
        self.stage = None
        self.calls = 0
        self.elements = 0

    def __call__(self, ptr, count, stream):
        torch = self.torch
        import torch.distributed as dist
        # zero-copy view of the library's device buffer (the solver stream has been synchronised by the caller)
        try:
            view = torch.as_tensor(_DevicePointer(ptr, count), device=self.device)
        except Exception:
            view = None
        if view is not None and view.data_ptr() == ptr:
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(self.device)
        else:       # staging fallback
            if self.stage is None or self.stage.numel() < count:
                self.stage = torch.empty(max(count, 1 << 16), dtype=torch.float64, device=self.device)
            t = self.stage[:count]
            nbytes = count * 8
            if self.hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(ptr), nbytes, hipMemcpyDeviceToDevice) != 0:
                raise RuntimeError("hipMemcpy to staging failed")
            self.hip.hipDeviceSynchronize()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(self.device)
            if self.hip.hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(t.data_ptr()), nbytes, hipMemcpyDeviceToDevice) != 0:
                raise RuntimeError("hipMemcpy from staging failed")
            self.hip.hipDeviceSynchronize()
        self.calls += 1
        self.elements += count


class _DevicePointer:
    """Minimal __cuda_array_interface__ carrier: lets torch alias `count` doubles at a raw device address."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2, "strides": None}


class GlooAllReduce:
    """Host-staged SUM all-reduce over a gloo group.  `host=True` treats the pointer as host memory
    (CPU tests of the hook itself); otherwise it is a device pointer staged through the host."""

    def __init__(self, host=False, group=None):
        self.host = host
        self.group = group
        self.hip = None if host else _hip()
        self.calls = 0

    def __call__(self, ptr, count, stream):
        import torch
        import torch.distributed as dist
        buf = np.empty(count, dtype=np.float64)
        if self.host:
            ctypes.memmove(buf.ctypes.data, ptr, count * 8)
        else:
            if self.hip.hipMemcpy(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(ptr), count * 8, hipMemcpyDeviceToHost) != 0:
                raise RuntimeError("hipMemcpy D2H failed")
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        if self.host:
            ctypes.memmove(ptr, buf.ctypes.data, count * 8)
        else:
            if self.hip.hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(buf.ctypes.data), count * 8, hipMemcpyHostToDevice) != 0:
                raise RuntimeError("hipMemcpy H2D failed")
        self.calls += 1


def init_rccl_comm(rank, world_size, device_index):
    """Bootstrap the library's own RCCL communicator: rank 0 creates the unique id and broadcasts it over the
    (already initialised) torch.distributed default group; every rank then joins on its device."""
    import torch.distributed as dist
    from .chain_bundle import Comm, comm_unique_id
    uid, err = None, None
    if rank == 0:
        try:
            uid = comm_unique_id()
        except Exception as exc:          # (the other ranks are waiting in the broadcast: tell them instead of leaving them there)
            err = repr(exc)
    box = [uid, err]
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        raise RuntimeError("rank 0 could not create the RCCL unique id: %s" % (box[1],))
    return Comm(box[0], rank, world_size, device_index)
