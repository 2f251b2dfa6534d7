"""N>1 path on CPU: world_size-2 gloo processes, each solving its contiguous shard with the CPU
oracle standing in for the local GPU solver (test infrastructure), the status all-reduce folding the
global solve status exactly as bench.py / ShardedSolver do over RCCL on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from tinympc_julia_amd.sharding import shard_range
    for batch in (1, 7, 64, 65536, 2**20, 1000003):
        for world in (1, 2, 3, 4, 8):
            got, prev = [], 0
            for r in range(world):
                lo, hi = shard_range(batch, r, world)
                assert lo == prev and hi >= lo
                prev = hi
                got.append(hi - lo)
            assert prev == batch and max(got) - min(got) <= 1
    assert shard_range(2**20, 3, 8) == (3 * 2**17, 4 * 2**17)   # config 5: 2^17 contiguous instances each
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_decode_status_roundtrip():
    from tinympc_julia_amd.sharding import decode_status
    res = np.array([1e-4, 2.5, 0.0, 3e-7], dtype=np.float32)
    words = np.zeros(8, dtype=np.uint32)
    words[:4] = res.view(np.uint32)
    assert decode_status(words)[0] == 0 and np.array_equal(decode_status(words)[1], res)
    words[4] = 3
    assert decode_status(words)[0] == 1
    # non-negative floats order like their bit patterns: integer MAX == float max
    a, b = np.float32(0.3).view(np.uint32), np.float32(0.7).view(np.uint32)
    assert (a < b) and np.float32(1e-30).view(np.uint32) < np.float32(1e-3).view(np.uint32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, tol, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tinympc_julia_amd as t
    from oracle import cpu_oracle
    from tinympc_julia_amd.sharding import ShardedSolver

    prob = t.problems.cartpole(20, u_bound=0.5)
    x0_all = t.problems.cartpole_x0(total, seed=0)

    class OracleLocal:
        """stands in for the rank's GPU BatchSolver: same status-block contract"""

        def __init__(self, n, lo, hi):
            self.x0 = np.asfortranarray(x0_all[:, lo:hi])
            self.st = torch.zeros(8, dtype=torch.int32)

        def solve_async(self):
            r = cpu_oracle.solve_batch("orc64", prob, self.x0, abs_pri_tol=tol, abs_dua_tol=tol, max_iter=40)
            self.result = r
            w = np.zeros(8, dtype=np.uint32)
            w[:4] = r["res"].max(axis=0).astype(np.float32).view(np.uint32)
            w[4] = int((r["solved"] == 0).sum())
            self.st = torch.from_numpy(w.view(np.int32).copy())

        def status_tensor(self):
            return self.st

        def synchronize(self):
            pass

    ss = ShardedSolver(lambda n, lo, hi: OracleLocal(n, lo, hi), total)
    status, res = ss.solve()
    # pipelined form (bench.py): back-to-back solves, each status exchange started on a snapshot and awaited later,
    # while the live block is already being overwritten by the next solve
    from tinympc_julia_amd.sharding import allreduce_status_async, decode_status
    pend = []
    for _ in range(3):
        ss.local.solve_async()
        pend.append(allreduce_status_async(ss.local.status_tensor()))
        ss.local.st.zero_()                      # what the next solve does to the live block
    for _, work in pend:
        work.wait()
    pipe = [decode_status(snap.numpy()) for snap, _ in pend]
    assert all(p[0] == status and np.array_equal(p[1], res) for p in pipe)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), status=status, res=res, lo=ss.lo, hi=ss.hi,
             iters=ss.local.result["iter"], u=ss.local.result["u"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tol,expect_status", [(5.0, 0), (1e-7, 1)])
def test_two_rank_sharded_solve_gloo(tmp_path, oracle_built, tol, expect_status):
    world, total = 2, 37   # ragged: 19 + 18
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, tol, str(tmp_path)), nprocs=world, join=True)
    import tinympc_julia_amd as t
    prob = t.problems.cartpole(20, u_bound=0.5)
    x0 = t.problems.cartpole_x0(total, seed=0)
    ref = oracle_built.solve_batch("orc64", prob, x0, abs_pri_tol=tol, abs_dua_tol=tol, max_iter=40)
    outs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    assert [(int(o["lo"]), int(o["hi"])) for o in outs] == [(0, 19), (19, 37)]
    want_res = ref["res"].max(axis=0).astype(np.float32)
    for o in outs:
        # every rank sees the same GLOBAL status and residual maxima after the all-reduce
        assert int(o["status"]) == expect_status == int(np.any(ref["solved"] == 0))
        assert np.array_equal(o["res"], want_res)
    # shards tile the batch: concatenated local results == the single-process solve
    assert np.array_equal(np.concatenate([o["iters"] for o in outs]), ref["iter"])
    assert np.array_equal(np.concatenate([o["u"] for o in outs], axis=2), ref["u"])
