"""Multi-GPU entry of the C-ABI (include/tinympc_hip.h section 3; SURVEY.md 8(b) last row, 8(e)).

CPU part: the partition rule and the error conventions (no compute).  GPU part (one-GPU box): the whole multi-shard
path for real — per-shard solvers and streams, scatter / gather by offset, the status fold — with (a) one shard, folded
by a one-rank RCCL communicator (librccl loaded, ncclCommInitAll, ncclAllReduce on the shard's stream), and (b) several
shards placed on the same device, which RCCL cannot connect, folded on the host.  Results must be BIT-identical to the
single-device BatchSolver on the same batch: sharding only moves instances."""
import ctypes
import os

import numpy as np
import pytest

import tinympc_julia_amd as t
from tinympc_julia_amd import sharding


def test_shard_range_matches_python_rule(hip_lib):
    for batch in (1, 2, 7, 37, 64, 65536, 2 ** 20, 1000003):
        for n in (1, 2, 3, 4, 8):
            if n > batch:
                continue
            prev = 0
            for r in range(n):
                lo, hi = t.shard_range(batch, n, r)
                assert (lo, hi) == sharding.shard_range(batch, r, n) and lo == prev
                prev = hi
            assert prev == batch
    assert t.shard_range(2 ** 20, 8, 3) == (3 * 2 ** 17, 4 * 2 ** 17)      # config 5


def test_sharded_without_gpu_fails_loudly(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = t.problems.cartpole(20, u_bound=0.5)
    with pytest.raises(t.TinyMPCError, match="no HIP device"):
        t.ShardedBatchSolver(p.A, p.B, p.Q, p.R, p.rho, p.N, batch=8, n_gpus=2)
    assert hip_lib.set_gpus(2) == -1 and b"not initialized" in hip_lib.tinympc_last_error()
    assert hip_lib.get_gpus() == 0


def _configure(bs, prob, x0, tol, max_iter, check=1):
    bs.update_settings(abs_pri_tol=tol, abs_dua_tol=tol, max_iter=max_iter, check_termination=check)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_x0(x0)


def _n_devices():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise the GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,backend", [([0], "rccl"), ([0, 0, 0], "host"), ([0, 1], "rccl")])
@pytest.mark.parametrize("tol,max_iter", [(0.0, 60), (1e-3, 25)])
def test_sharded_equals_single_device(hip_lib, devices, backend, tol, max_iter):
    if max(devices) >= _n_devices():
        # two DISTINCT devices: ncclCommInitAll across real devices and the grouped ncclAllReduce over xGMI — runs on the
        # first multi-GPU lease (the one-GPU test box has no second device)
        pytest.skip(f"needs {max(devices) + 1} GPUs, {_n_devices()} present")
    prob = t.problems.cartpole(20, u_bound=0.5)
    B = 37                                                    # ragged over 3 shards: 13 + 12 + 12
    x0 = t.problems.cartpole_x0(B, seed=5)
    x0[:, 7] *= 6.0                                           # one hard instance: unsolved within 25 iterations
    one = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, device=0)
    _configure(one, prob, x0, tol, max_iter)
    st1 = one.solve()
    s1, q1 = one.get_solution(), one.get_status()
    sh = t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, devices=devices)
    assert sh.fold_backend == backend and sh.n_shards == len(devices)
    assert [sh.shard(i)[1:3] for i in range(sh.n_shards)] == [sharding.shard_range(B, i, len(devices))
                                                              for i in range(len(devices))]
    _configure(sh, prob, x0, tol, max_iter)
    for rep in range(2):                                      # warm-started second solve goes through the same plumbing
        st = sh.solve()
        s, q = sh.get_solution(), sh.get_status()
        if rep == 0:
            assert st == st1 == int(np.any(q1["solved"] == 0))
            assert np.array_equal(q["iter"], q1["iter"]) and np.array_equal(q["solved"], q1["solved"])
            assert np.array_equal(s["states"], s1["states"]) and np.array_equal(s["controls"], s1["controls"])
            assert np.array_equal(q["residuals"], q1["residuals"])
            res, unsolved = sh.global_status()
            assert np.array_equal(res.astype(np.float32), q1["residuals"].max(axis=0).astype(np.float32))
            assert (unsolved > 0) == (st == 1)
            ws1 = one.get_workspace()
            ws = sh.get_workspace()
            assert all(np.array_equal(ws[k], ws1[k]) for k in ws)
            one.solve()                                       # the single-device solver's second (warm-started) solve
    s1b, sb = one.get_solution(), sh.get_solution()
    assert np.array_equal(sb["controls"], s1b["controls"])
    # async form: enqueue, then wait
    sh.reset()
    sh.solve_async()
    assert sh.wait() == st1
    assert np.array_equal(sh.get_solution()["controls"], s1["controls"])
    sh.close()
    one.close()


@pytest.mark.gpu
def test_sharded_per_instance_refs_and_quadrotor_kernel(hip_lib):
    """per-instance references are scattered by offset too; each shard picks its own kernel for its own batch"""
    prob = t.problems.quadrotor(30, u_bound=0.5)
    B = 70
    rng = np.random.default_rng(9)
    x0 = t.problems.quadrotor_x0(B, seed=4)
    xr = 0.05 * rng.standard_normal((12, 30, B))
    ur = 0.02 * rng.standard_normal((4, 29, B))
    outs = []
    for mk in (lambda: t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, device=0),
               lambda: t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, devices=[0, 0])):
        bs = mk()
        _configure(bs, prob, x0, 0.0, 30)
        bs.set_x_ref(xr)
        bs.set_u_ref(ur)
        bs.solve()
        outs.append(bs.get_solution())
        bs.close()
    assert np.array_equal(outs[0]["states"], outs[1]["states"]) and np.array_equal(outs[0]["controls"], outs[1]["controls"])


@pytest.mark.gpu
def test_global_solver_set_gpus(hip_lib, monkeypatch):
    """the drop-in surface: setup(batch) -> set_gpus(n) -> every bare entry point acts on the whole sharded batch"""
    prob = t.problems.cartpole(20, u_bound=0.5)
    B = 23
    x0 = t.problems.cartpole_x0(B, seed=11)

    def run(n_gpus):
        s = t.TinyMPCSolver()
        t.setup(s, prob.A, prob.B, None, prob.Q, prob.R, prob.rho, 4, 1, 20, batch=B, n_gpus=n_gpus, max_iter=40,
                abs_pri_tol=1e-3, abs_dua_tol=1e-3)
        assert hip_lib.get_gpus() == n_gpus and hip_lib.get_batch_size() == B
        t.set_bound_constraints(s, prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        t.set_x0(s, x0)
        st = t.solve(s)
        return st, t.get_solution(s), t.get_status(s)

    st1, sol1, q1 = run(1)
    monkeypatch.setenv("TINYMPC_HIP_SHARD_DEVICES", "0,0")
    st2, sol2, q2 = run(2)
    assert st1 == st2 and np.array_equal(q1["iter"], q2["iter"])
    assert np.array_equal(sol1["states"], sol2["states"]) and np.array_equal(sol1["controls"], sol2["controls"])
    # re-batching keeps the device count; n = 1 returns to the single device
    s = t.TinyMPCSolver()
    t.setup(s, prob.A, prob.B, None, prob.Q, prob.R, prob.rho, 4, 1, 20, batch=B, n_gpus=2)
    t.set_batch_size(s, 9)
    assert hip_lib.get_gpus() == 2 and hip_lib.get_batch_size() == 9
    t.set_gpus(s, 1)
    assert hip_lib.get_gpus() == 1 and hip_lib.get_batch_size() == 9
    monkeypatch.delenv("TINYMPC_HIP_SHARD_DEVICES")
    with pytest.raises(t.TinyMPCError, match="does not exist"):
        t.set_gpus(s, 9)                                         # more devices than the box has
    assert hip_lib.get_gpus() in (0, 1)
    t.cleanup()


@pytest.mark.gpu
def test_sharded_persistent_and_refill_kernels_share_a_device(hip_lib):
    """two shards on ONE device launch their kernels concurrently on their own streams: the persistent on-chip kernel
    (rocket, cones + affine term set through each shard's own handle, tinympc_sharded_shard) and the matrix-core kernel's
    refill launch (tolerance-terminated quadrotor, enough instances per shard for two rounds) each size their grid for the
    whole chip and take work off their own counters — results must equal the single-device solver's, bit for bit"""
    import ctypes
    lib = hip_lib
    # (a) rocket N = 50, cones + fdyn, one-shot: mfmat on both shards
    N, B = 50, 96
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=4)
    xr, ur = t.problems.rocket_refs(N)
    kw = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=1)
    one = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, device=0)
    one.update_settings(**kw)
    one.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    one.set_fdyn(prob.fdyn)
    one.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
    one.set_warm_start(False); one.set_x_ref(xr); one.set_u_ref(ur); one.set_x0(x0)
    one.solve()
    ref = one.get_solution()
    assert one.kernel_name == "mfmat<6,3,50>"
    one.close()
    sh = t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, devices=[0, 0])
    _configure(sh, prob, x0, 0.0, 60)
    fd = np.ascontiguousarray(prob.fdyn, dtype=np.float64)
    ia = lambda v: np.ascontiguousarray(v, dtype=np.int32).ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    da = lambda v: np.ascontiguousarray(v, dtype=np.float64).ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    for i in range(sh.n_shards):
        loc = sh.shard(i)[3]
        assert lib.tinympc_set_fdyn(loc, fd.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
        assert lib.tinympc_set_cone_constraints(loc, ia([0]), ia([3]), da([0.25]), 1, ia([0]), ia([3]), da([0.5]), 1) == 0
    sh.set_warm_start(False); sh.set_x_ref(xr); sh.set_u_ref(ur)
    sh.solve()
    assert sh.kernel_names() == ["mfmat<6,3,50>"] * 2
    got = sh.get_solution()
    assert np.array_equal(got["states"], ref["states"]) and np.array_equal(got["controls"], ref["controls"])
    sh.close()
    # (b) quadrotor, tolerance-terminated, 2 x 40 960 instances: the refill launch on both shards at once
    prob = t.problems.quadrotor(30, u_bound=0.5)
    B = 81920
    x0 = t.problems.quadrotor_x0(B, seed=13)
    x0[:, ::3] *= 0.1
    outs = []
    for mk in (lambda: t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, device=0),
               lambda: t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, devices=[0, 0])):
        bs = mk()
        bs.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=10)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        bs.set_warm_start(False)
        bs.set_x0(x0)
        st = bs.solve()
        outs.append((st, bs.get_solution(), bs.get_status()))
        bs.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][2]["iter"], outs[1][2]["iter"]) and len(np.unique(outs[0][2]["iter"])) > 3
    assert np.array_equal(outs[0][1]["controls"], outs[1][1]["controls"])
    assert np.array_equal(outs[0][1]["states"], outs[1][1]["states"])


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0, 0], [0, 1]])
def test_sharded_chunked_solves_run_side_by_side(hip_lib, devices):
    """a solve in chunks with compaction synchronises its stream on the host between chunks: the sharded handle gives each
    shard a host thread (enqueued one after another, shard i would finish before shard i + 1 starts).  Results are the
    single-device chunked solve's; on two real devices the wall time is about one shard's, not two."""
    import time
    if max(devices) >= _n_devices():
        pytest.skip(f"needs {max(devices) + 1} GPUs, {_n_devices()} present")
    prob = t.problems.quadrotor(30, u_bound=0.5)
    B = 32768
    x0 = t.problems.quadrotor_x0(B, seed=17)
    x0[:, ::3] *= 0.1
    outs, secs = [], []
    for mk in (lambda: t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, device=0),
               lambda: t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, devices=devices)):
        bs = mk()
        _configure(bs, prob, x0, 1e-3, 100, check=10)
        bs.set_compaction(20)
        bs.solve()
        bs.reset()
        t0 = time.perf_counter()
        st = bs.solve()
        secs.append(time.perf_counter() - t0)
        outs.append((st, bs.get_solution(), bs.get_status()))
        bs.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][2]["iter"], outs[1][2]["iter"]) and len(np.unique(outs[0][2]["iter"])) > 3
    assert np.array_equal(outs[0][1]["controls"], outs[1][1]["controls"])
    if devices[0] != devices[1]:
        assert secs[1] <= 0.85 * secs[0], secs             # two devices, half the batch each, side by side


@pytest.mark.gpu
def test_config5_full_size_eight_shards_one_device(hip_lib, oracle_built):
    """BASELINE config 5 at its stated size — quadrotor (12,4,30), 2^20 instances (seed 3), tolerance 1e-3 checked every
    10 iterations — through the in-process multi-GPU handle (tinympc_create_sharded, what a Julia host drives) with the
    eight shards of 2^17 instances placed on the one device of this box (host fold).  Full-size properties: determinism
    (two cold solves, bit-equal), feasibility, iteration counts on the check grid, the status fold equal to the
    per-instance status; the instances on either side of every shard boundary bit-equal to a single-solver run; and
    parity with the fp64 oracle on a stratified sample of 4 096 instances (every 256th)."""
    from tests.util import parity_every_instance
    B, S = 2 ** 20, 8
    prob, x0 = t.problems.quadrotor(30), t.problems.quadrotor_x0(B, seed=3)
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=10)
    sh = t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, devices=[0] * S)
    assert sh.n_shards == S and sh.fold_backend == "host"
    assert [sh.shard(i)[1:3] for i in range(S)] == [(i * 2 ** 17, (i + 1) * 2 ** 17) for i in range(S)]
    sh.update_settings(**kw)
    sh.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    sh.set_warm_start(False)
    sh.set_x0(x0)
    status = sh.solve()
    sol, st = sh.get_solution(), sh.get_status()
    assert set(sh.kernel_names()) == {"mfma<12,4,30>"}
    # full-size properties
    assert np.isfinite(sol["states"]).all() and np.isfinite(sol["controls"]).all()
    assert (sol["controls"] <= prob.u_max[:, :, None]).all() and (sol["controls"] >= prob.u_min[:, :, None]).all()
    assert np.all(st["iter"] % 10 == 0) and st["iter"].min() >= 10 and st["iter"].max() == 100
    assert np.all((st["solved"] == 1) | (st["iter"] == 100))
    assert status == int(np.any(st["solved"] == 0))
    res, unsolved = sh.global_status()
    assert np.array_equal(res.astype(np.float32), st["residuals"].max(axis=0).astype(np.float32))
    per_shard_unsolved = [(st["solved"][i * 2 ** 17:(i + 1) * 2 ** 17] == 0).sum() for i in range(S)]
    assert unsolved == max(per_shard_unsolved)
    u_first, it_first = sol["controls"].copy(), st["iter"].copy()
    # determinism: a second cold solve of the same batch
    sh.reset()
    assert sh.solve() == status
    again = sh.get_solution()["controls"]
    assert np.array_equal(again, u_first) and np.array_equal(sh.get_status()["iter"], it_first)
    del again
    sh.close()
    # the instances on either side of every shard boundary (and the batch's ends) on ONE plain solver
    edge = np.concatenate([np.arange(0, 64)] + [np.arange(k * 2 ** 17 - 64, k * 2 ** 17 + 64) for k in range(1, S)] + [np.arange(B - 64, B)])
    one = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=len(edge))
    one.update_settings(**kw)
    one.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    one.set_warm_start(False)
    one.set_x0(np.asfortranarray(x0[:, edge]))
    one.solve()
    s1, q1 = one.get_solution(), one.get_status()
    assert np.array_equal(s1["controls"], u_first[:, :, edge]) and np.array_equal(s1["states"], sol["states"][:, :, edge])
    assert np.array_equal(q1["iter"], it_first[edge])
    one.close()
    # oracle parity on a stratified sample
    pick = np.arange(0, B, 256)
    xs = np.asfortranarray(x0[:, pick])
    ref = oracle_built.solve_batch("orc64", prob, xs, nthreads=len(os.sched_getaffinity(0)), **kw)

    def make(b=None):
        o = oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        return o
    sub = dict(states=sol["states"][:, :, pick], controls=sol["controls"][:, :, pick])
    frac = parity_every_instance(sub, {k: v[pick] for k, v in st.items()}, ref, make, xs, kw, prob.rho, min_same=0.99, tag="config 5, 2^20")
    print(f"config 5 at 2^20 over 8 shards: mean iterations {st['iter'].mean():.1f}, unsolved {int((st['solved'] == 0).sum())}, "
          f"sample of {len(pick)}: {frac:.4f} of the iteration counts agree with the oracle")
