"""RCCL called directly (ctypes), for the ONE collective on the env's path: the per-step all-gather of the kernel-filled block.

OPT-IN (`ShardedEnv(direct=True)`, `bench.py --direct-rccl`).  Why it exists: torch.distributed's own `all_gather_into_tensor` costs 24-38 us of HOST
time per call on this stack (ProcessGroupNCCL's bookkeeping; measured with the RCCL backend and the one rank a one-GPU box allows,
profiles/r06_k_gather_host_cost.txt, r06_l_gather_host_cost.txt: 29-43 us per overlapped step enqueued against 5 without a collective) -- several times
the 11-us step kernel it is meant to run under, so a per-step gather through it makes an N-GPU job host-bound.  `ncclAllGather` called directly, with an
event pair around it (the comm stream waits for the step kernel, the caller's stream waits for the gather when IT wants to): 23 us per step on the same
box -- RCCL's own enqueue is most of what is left.  Why it is not the default: its multi-rank initialisation (unique id through the process group,
ncclCommInitRank on every rank) has never run on real hardware -- RCCL refuses two ranks on one device and the builder never had a multi-GPU node; the
one-rank form is tested (tests/test_api_gpu.py::test_rccl_backend_runs_the_gather_path_on_one_rank[True]).

torch.distributed remains the control plane: the communicator's unique id travels through the existing process group (any backend), and so does the
verdict of the self-check (an all-gather of rank ids through the new communicator, compared on every rank, agreed on by an all-reduce) -- a
communicator that does not pass is not used, and `ShardedEnv` falls back to the process group's collective.  The library is the librccl.so torch itself
loaded (torch/lib), so both paths drive the same RCCL build.  Nothing here has a reference counterpart (the reference is single-process).
"""
import ctypes as C
import os

import torch

NCCL_UNIQUE_ID_BYTES = 128   # rccl.h:40
NCCL_FLOAT32, NCCL_INT32 = 7, 2   # rccl.h ncclDataType_t


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]   # rccl.h:43


_lib = None


def _rccl():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = C.CDLL(path if os.path.exists(path) else "librccl.so")
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]   # (the id travels BY VALUE)
        lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllGather, lib.ncclCommDestroy):
            f.restype = C.c_int
        _lib = lib
    return _lib


class RcclError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise RcclError(f"{what}: {_rccl().ncclGetErrorString(rc).decode()} ({rc})")


class _Pending:
    """handle of one enqueued gather: wait() orders the CURRENT stream behind it (no host synchronisation)"""

    def __init__(self, event):
        self.event = event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
            self.event = None
        return True


class DirectComm:
    """One RCCL communicator over the ranks of a torch.distributed group, on its own HIP stream of this rank's device."""

    def __init__(self, rank, world_size, device, group=None):
        import torch.distributed as dist
        self.rank, self.world_size, self.device = int(rank), int(world_size), torch.device(device)
        self._comm = C.c_void_p()
        lib = _rccl()
        uid = _UniqueId()
        rc0 = lib.ncclGetUniqueId(C.byref(uid)) if self.rank == 0 else 0
        raw = bytes(C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES)) if (self.rank == 0 and rc0 == 0) else bytes(NCCL_UNIQUE_ID_BYTES)
        if self.world_size > 1:   # the id travels through the process group that already works (a uint8 tensor where that group's tensors live);
            # EVERY rank reaches the broadcast, whatever rank 0's call returned: an all-zero id is the "failed" message, on which every rank raises below
            where = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
            t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone().to(where)
            dist.broadcast(t, src=0, group=group)
            raw = t.cpu().numpy().tobytes()
        if not any(raw):
            raise RcclError(f"ncclGetUniqueId failed on rank 0 ({rc0})" if self.rank == 0 else "rank 0 could not create a unique id")
        C.memmove(C.addressof(uid), raw, NCCL_UNIQUE_ID_BYTES)
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(C.byref(self._comm), self.world_size, uid, self.rank), "ncclCommInitRank")
            self.stream = torch.cuda.Stream(device=self.device)
        self._ready = [torch.cuda.Event() for _ in range(2)]
        self._done = [torch.cuda.Event() for _ in range(2)]
        self._k = 0

    def all_gather(self, out, block):
        """enqueue out[world_size * m, row] <- every rank's block[m, row], behind everything already on the CURRENT stream; -> _Pending"""
        if not (out.is_contiguous() and block.is_contiguous() and out.dtype == block.dtype == torch.float32 and out.numel() == self.world_size * block.numel()):
            raise RcclError("all_gather: contiguous fp32 tensors with out = world_size x block expected")
        k = self._k
        self._k ^= 1
        self._ready[k].record(torch.cuda.current_stream())
        self.stream.wait_event(self._ready[k])
        _check(_rccl().ncclAllGather(block.data_ptr(), out.data_ptr(), block.numel(), NCCL_FLOAT32, self._comm, C.c_void_p(self.stream.cuda_stream)), "ncclAllGather")
        self._done[k].record(self.stream)
        return _Pending(self._done[k])

    def self_check(self):
        """this rank's part: an all-gather of rank ids through THIS communicator equals [0 .. world_size); never raises"""
        try:
            mine = torch.full((1, 32), float(self.rank), dtype=torch.float32, device=self.device)
            out = torch.full((self.world_size, 32), -1.0, dtype=torch.float32, device=self.device)
            self.all_gather(out, mine).wait()
            torch.cuda.current_stream().synchronize()
            return bool(torch.equal(out[:, 0].cpu(), torch.arange(self.world_size, dtype=torch.float32))), None
        except Exception as e:  # noqa: BLE001
            return False, repr(e)[:200]

    def destroy(self):
        if self._comm:
            try:
                self.stream.synchronize()
                _rccl().ncclCommDestroy(self._comm)
            finally:
                self._comm = C.c_void_p()


_cache = {}


def _agree(ok, world_size, device, group):
    """every rank learns whether ALL ranks said ok (through the process group; every rank reaches this call whatever happened before it)"""
    if world_size <= 1:
        return ok
    import torch.distributed as dist
    where = torch.device(device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flag = torch.tensor([1.0 if ok else 0.0], device=where)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() == 1.0)


def make_direct_comm(rank, world_size, device, group=None):
    """-> (DirectComm or None, reason).  Every rank calls this at the same point of its program; a communicator that cannot be created or does not
    pass its self-check on EVERY rank is used on none of them.  One communicator per (group, world size, rank, device) and process."""
    key = (id(group) if group is not None else 0, int(world_size), int(rank), str(device))
    if key in _cache:
        return _cache[key]
    comm, why = None, None
    try:
        comm = DirectComm(rank, world_size, device, group)
    except Exception as e:  # noqa: BLE001
        why = repr(e)[:200]
    if not _agree(comm is not None, world_size, device, group):   # (no rank enqueues a collective on a communicator some rank does not have)
        if comm is not None:
            comm.destroy()
        _cache[key] = (None, why or "another rank could not create its communicator")
        return _cache[key]
    ok, err = comm.self_check()
    if not _agree(ok, world_size, device, group):
        comm.destroy()
        _cache[key] = (None, err or why or "the communicator's self-check (an all-gather of rank ids) failed on some rank")
        return _cache[key]
    _cache[key] = (comm, None)
    return _cache[key]
