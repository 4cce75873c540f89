"""An own RCCL communicator, driven through the library's C API with ctypes: collectives issued ON THE CALLER'S STREAM.

Why not torch.distributed's NCCL backend for the per-mini-epoch exchanges: ProcessGroupNCCL runs every collective on a stream of its own between two
events (record on the caller's stream, wait on the group's, collective, record, wait back).  Measured in a world of one rank (round 4,
profiles/r04_timeline_mini_epoch_world1_rccl.txt): +54 us per mini-epoch of pure stream hand-over around collectives whose own kernels take 4 us -- what
every rank of a multi-GPU job pays before a byte has crossed xGMI.  Here `ncclAllReduce(..., stream)` is enqueued on the stream the producer kernel
was launched on, like any other launch of the update; several exchanges between ncclGroupStart / ncclGroupEnd become ONE launch.

The process group of torch.distributed stays what it was for everything that is not per-mini-epoch (rendezvous, the broadcast of the initial weights, the
curriculum grid, barriers); it also carries this communicator's 128-byte unique id from rank 0 to the others.  The library is the one torch itself
loaded (torch/lib/librccl.so): one RCCL in the process.  Replaces the collective that sits where the reference clips and steps
(utils/runner.py:162-165 on one process); SURVEY section 8(e).
"""
import ctypes as C
import os

import torch

NCCL_SUM, NCCL_MAX, NCCL_AVG = 0, 2, 4
_DTYPES = {torch.float32: 7, torch.float64: 8, torch.int32: 2, torch.int64: 4}


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path)
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy, lib.ncclGroupStart, lib.ncclGroupEnd):
        f.restype = C.c_int
    return lib


class RcclComm:
    """One communicator over all ranks of the job, brought up in two LOCAL steps with the ranks' agreement in between (utils/parallel.py:
    own_comm_bring_up): `prepare` (load the library; rank 0 draws the unique id) cannot block and may raise; `RcclComm(...)` = ncclCommInitRank blocks
    until every rank has called it and is therefore entered only when every rank has reported a successful `prepare` and holds the id."""

    @staticmethod
    def prepare(rank):
        """-> (library handle, rank 0: the 128-byte unique id, other ranks: None).  Raises where the library or its entry points are missing."""
        lib = _load()
        if rank != 0:
            return lib, None
        uid = _UniqueId()
        rc = lib.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId failed ({rc}): {lib.ncclGetErrorString(rc).decode()}")
        return lib, C.string_at(C.byref(uid), 128)  # (not bytes(uid.internal): a c_char array stops at the first NUL)

    def __init__(self, lib, raw_id, rank, world_size, device_index):
        self.lib = lib
        self.rank, self.world_size = rank, world_size
        if raw_id is None or len(raw_id) != 128:
            raise RuntimeError("RCCL unique id: expected 128 bytes")
        uid = _UniqueId()
        C.memmove(C.byref(uid), raw_id, 128)
        torch.cuda.set_device(device_index)
        self.comm = C.c_void_p()
        self._ok(self.lib.ncclCommInitRank(C.byref(self.comm), world_size, uid, rank), "ncclCommInitRank")

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.ncclGetErrorString(rc).decode()}")

    def all_reduce_(self, t, op=NCCL_SUM):
        """In place, on torch's current stream."""
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("RcclComm.all_reduce_: contiguous device tensor expected")
        p = t.data_ptr()
        self._ok(self.lib.ncclAllReduce(p, p, t.numel(), _DTYPES[t.dtype], op, self.comm, torch.cuda.current_stream().cuda_stream), "ncclAllReduce")
        return t

    def group(self):
        return _Group(self)

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


class _Group:
    """with comm.group(): several all_reduce_ calls -> one launch."""

    def __init__(self, comm):
        self.c = comm

    def __enter__(self):
        self.c._ok(self.c.lib.ncclGroupStart(), "ncclGroupStart")
        return self.c

    def __exit__(self, *exc):
        self.c._ok(self.c.lib.ncclGroupEnd(), "ncclGroupEnd")
        return False
