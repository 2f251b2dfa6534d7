"""Multi-GPU: one process per GPU, the batch sharded in contiguous ranges, no data-path collective.

Instances are independent, so rank r of W solves instances
[r*B/W, (r+1)*B/W) entirely on its own GPU; x0/refs go in and solutions stay sharded.  The
path's only exchange is the solve status: `solve_mpc` returns 0 iff EVERY instance converged
and reports the global residual maxima, so each solve ends with one all-reduce(MAX) of the
5-word status block over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  20 bytes: latency-bound, independent of link bandwidth.

Status block words (uint32, admm_params.h): [0..3] float bits of max pri_x, dua_x, pri_u, dua_u
(non-negative floats order like integers, so an integer MAX is a float max), [4] number of local
instances that hit max_iter (MAX over ranks: > 0 iff any instance anywhere is unsolved).
"""
import numpy as np

STATUS_WORDS = 8


def shard_range(batch, rank, world):
    """Contiguous shard [lo, hi) of `batch` instances for `rank` of `world` (sizes differ by <= 1)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _DevPtr:
    """Zero-copy view of library-owned device memory for torch (__cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


def device_tensor(ptr, shape, dtype, device):
    """torch tensor aliasing a device buffer returned by tinympc_device_buffers (no copy)."""
    import torch
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.float64: "<f8"}[dtype]
    return torch.as_tensor(_DevPtr(ptr, shape, typestr), device=device)


def allreduce_status(status, group=None):
    """In-place all-reduce(MAX) of a status block (int32 tensor, CPU or GPU).  The collective is
    enqueued on the tensor's current stream on GPU (RCCL); blocking on CPU (gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(status, op=dist.ReduceOp.MAX, group=group)
    return status


def allreduce_status_async(status, group=None, force=False):
    """Pipelined form for back-to-back solves: snapshots the status block (stream-ordered after the solve that
    produced it) and starts the all-reduce(MAX) of the snapshot without making the launch stream wait for it, so
    the exchange overlaps the next solve (which zeroes and rewrites the live block).  Returns (snapshot, work);
    `work.wait()` (None on a single rank) must be called before the snapshot is read."""
    import torch.distributed as dist
    snap = status.clone()
    work = None
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        work = dist.all_reduce(snap, op=dist.ReduceOp.MAX, group=group, async_op=True)
    return snap, work


def decode_status(words):
    """status words -> (solve status 0/1, residual maxima float32[4])"""
    w = np.asarray(words).astype(np.uint32)
    res = w[:4].copy().view(np.float32)
    return (0 if int(w[4]) == 0 else 1), res


class ShardedSolver:
    """Drives one local shard per process and folds the status over the ranks.

    `make_local(n_local, lo, hi)` builds the rank's solver for instances [lo, hi); it must offer
      solve_async() / synchronize() / status_tensor()        (GPU: BatchSolver-backed, see
    `local_from_batch_solver`), where status_tensor() is the int32[8] status block of the last
    solve — a CUDA tensor aliasing the library's device block on the GPU path.
    """

    def __init__(self, make_local, total_batch, group=None):
        import torch.distributed as dist
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self.total_batch = int(total_batch)
        self.lo, self.hi = shard_range(total_batch, self.rank, self.world)
        self.local = make_local(self.hi - self.lo, self.lo, self.hi)

    def solve(self):
        """One batched solve of every shard; returns (global status, global residual maxima)."""
        self.local.solve_async()
        st = self.local.status_tensor()
        allreduce_status(st, self.group)
        self.local.synchronize()
        return decode_status(st.detach().cpu().numpy())


def local_from_batch_solver(bs, device):
    """Adapter: a BatchSolver living on `device` as a ShardedSolver local shard (GPU path)."""
    import torch

    class _Local:
        def __init__(self):
            self.bs = bs
            self.dev = torch.device(device)
            self._st = device_tensor(bs.device_buffers()["gstat"], (STATUS_WORDS,), torch.int32, self.dev)

        def solve_async(self):
            self.bs.solve_async(torch.cuda.current_stream(self.dev).cuda_stream)

        def status_tensor(self):
            return self._st

        def synchronize(self):
            torch.cuda.synchronize(self.dev)

    return _Local()
