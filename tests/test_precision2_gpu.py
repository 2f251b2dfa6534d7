"""tinympc_set_precision(s, 2): the reference's arithmetic end to end — fp64 recurrences, slacks, duals, residual comparisons
and the workspace kept between solves (types.hpp:15) — on the generic kernel's double-state form ("generic<f64>").

It exists for the cases the fp32-state kernels do not hold at 1e-5 (round-3 review, "an fp64-state option"): the four
exception sites of the suite and the four seeds of the out-of-suite fuzz that missed are replayed here under precision 2
and must hold the PLAIN tolerance — no derived per-instance limits, no 2e-5 for the duals:
 * tests/test_gpu_parity.py:1337  quadrotor with state bounds that cannot be met (duals ~800 x the trajectory);
 * tests/test_gpu_parity.py:1666  warm-started closed loop with the workspace compared after every solve;
 * tests/test_mfmat_gpu.py:187    rocket with cones + affine term, workspace kept (converged-exit quirk included);
 * tests/test_mfmat_gpu.py:557    second warm solve of a random family with per-knot bounds;
 * profiles/r03_fuzz_large.txt    seeds 1109 (workspace) and 1125 (closed loop) of scripts/fuzz_mfmat.py.
Inputs (x0, references, bounds, diag(Q)+rho) still reach the kernel as the fp32 device arrays and the solution comes back
as fp32 arrays, so the floor is ~1e-7, not 1e-12; the bars below are 1e-6 where the oracle is fed the same fp32-rounded x0."""
import os
import sys

import numpy as np
import pytest

import tinympc_julia_amd as t
from tests.util import FP32_TOL, nrel, nrel_batch, parity_every_instance

pytestmark = pytest.mark.gpu
TIGHT = 1e-6


def _f32(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float32).astype(np.float64))


def _plain(oracle_built, prob, kw, xref=None, uref=None):
    def make(b=None):
        o = oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if xref is not None:
            o.set_x_ref(xref)
            o.set_u_ref(uref)
        return o
    return make


def test_unmeetable_state_bounds_hold_the_plain_tolerance(hip_lib, oracle_built):
    """test_matrix_core_kernel_vs_oracle[quadrotor30_refs_bounds] needs per-instance limits up to 6e-5 on the fp32-state kernels
    (state duals ~800 x the trajectory); with fp64 state every instance holds 1e-6"""
    rng = np.random.default_rng(5)
    N, B = 30, 171
    prob = t.problems.quadrotor(N)
    x0 = _f32(t.problems.quadrotor_x0(B, seed=4))
    prob.x_min, prob.x_max = np.full((12, N), -0.12), np.full((12, N), 0.12)
    prob.x_min[:, N // 2:] = -0.2
    xref, uref = _f32(0.05 * rng.standard_normal((12, N))), _f32(0.02 * rng.standard_normal((4, N - 1)))
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60, check_termination=1)
    ref = oracle_built.solve_batch("orc64", prob, x0, xref=xref, uref=uref, nthreads=len(os.sched_getaffinity(0)), **kw)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_warm_start(False)
    bs.set_x_ref(xref)
    bs.set_u_ref(uref)
    bs.set_precision(2)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name == "generic<f64>" and bs.effective_precision == 2
    sol, st = bs.get_solution(), bs.get_status()
    assert np.array_equal(st["iter"], ref["iter"]) and np.array_equal(st["solved"], ref["solved"])
    assert nrel_batch(sol["states"], ref["x"]).max() <= TIGHT and nrel_batch(sol["controls"], ref["u"]).max() <= TIGHT
    assert np.abs(st["residuals"] - ref["res"]).max() <= 1e-6 * max(1.0, np.abs(ref["res"]).max())
    # and the default precision on the same solver afterwards (the switch restarts the workspace cold)
    bs.set_precision(0)
    bs.solve()
    assert bs.kernel_name == "mfma<12,4,30>" and bs.effective_precision == 0
    bs.close()


@pytest.mark.parametrize("case", ["quadrotor_state_bounds", "rocket_cones_fdyn"])
def test_workspace_kept_closed_loop_in_fp64(hip_lib, oracle_built, case):
    """the reference's default calling pattern — solve() resets counters only, the workspace carries over (admm.cpp:111-115) —
    as a host-stepped closed loop with one persistent oracle per instance: solution, iteration count, solved flag and the
    workspace itself (d, y, g, v, z) after every solve, the duals at the same bar as everything else"""
    if case == "quadrotor_state_bounds":
        N, B, steps = 30, 24, 4
        prob = t.problems.quadrotor(N)
        prob.x_min, prob.x_max = np.full((12, N), -0.25), np.full((12, N), 0.25)
        x0 = _f32(t.problems.quadrotor_x0(B, seed=8))
        kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=40, check_termination=1)
        fdyn, cones, xr, ur = None, None, None, None
    else:
        N, B, steps = 10, 24, 5
        prob = t.problems.rocket(N)
        x0 = _f32(t.problems.rocket_x0(B, seed=5))
        kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
        fdyn, cones = prob.fdyn, ([0], [3], [0.25], [0], [3], [0.5])
        xr, ur = t.problems.rocket_refs(N)

    def configure(o):
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if fdyn is not None:
            o.set_fdyn(fdyn)
            o.set_cone_constraints(*cones)
            o.set_x_ref(xr)
            o.set_u_ref(ur)
        return o
    bs = configure(t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B))
    bs.set_warm_start(True)
    bs.set_precision(2)
    orcs = [configure(oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)) for _ in range(B)]
    f = fdyn if fdyn is not None else np.zeros(prob.nx)
    x = x0.copy()
    converged = 0
    for k in range(steps):
        bs.set_x0(x)
        bs.solve()
        assert bs.kernel_name == "generic<f64>"
        sol, st, ws = bs.get_solution(), bs.get_status(), bs.get_workspace()
        xn = np.zeros_like(x)
        for b in range(B):
            o = orcs[b]
            o.set_x0(x[:, b])
            o.solve()
            r = o.get_solution()
            assert st["iter"][b] == r["iter"] and st["solved"][b] == r["solved"], f"step {k} instance {b}: {st['iter'][b]} vs {r['iter']}"
            converged += r["solved"]
            assert nrel(sol["states"][:, :, b], r["x"]) <= TIGHT and nrel(sol["controls"][:, :, b], r["u"]) <= TIGHT
            sv = o.get_state()
            for key in ("d", "y", "g", "v", "z"):
                e_ = np.abs(ws[key][:, :, b] - sv[key]).max() / max(np.abs(sv[key]).max(), 1e-2)
                assert e_ <= TIGHT, f"step {k} instance {b} workspace {key}: {e_:.3e}"
            xn[:, b] = _f32(prob.A @ x[:, b] + prob.B @ r["u"][:, 0] + f)      # (what set_x0 hands the kernel: fp32)
        x = xn
    if case == "rocket_cones_fdyn":
        assert converged >= B // 2                              # converged exits (v, z one iteration old) are in the sample
    else:
        assert np.abs(bs.get_workspace()["g"]).max() > 1e-3     # the state bound binds: the state dual is live
    for o in orcs:
        o.close()
    bs.close()


def test_per_knot_bounds_random_family_second_warm_solve(hip_lib, oracle_built):
    """tests/test_mfmat_gpu.py:557 allows the second warm solve of this family 2e-5 on the fp32-state kernel"""
    rng = np.random.default_rng(150)
    nx, nu, N, B = 6, 3, 50, 45
    A = np.eye(nx) + 0.2 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= 0.96 / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)),
                              np.diag(rng.uniform(0.5, 3.0, nu)), float(rng.uniform(0.5, 2.0)), N)
    prob.x_min, prob.x_max = -rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N)), rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N))
    prob.x_min[3:, :], prob.x_max[3:, :] = -1e17, 1e17
    prob.x_min[:3, N // 2:] -= 0.3
    prob.u_min, prob.u_max = -rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1)), rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1))
    prob.u_max[:, ::2] += 0.1
    fdyn = 0.02 * rng.standard_normal(nx)
    xr, ur = 0.2 * rng.standard_normal((nx, N)), 0.1 * rng.standard_normal((nu, N - 1))
    cones = ([0], [3], [0.7], [0], [3], [1.1])
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60, check_termination=1)
    x0 = _f32(rng.uniform(-0.5, 0.5, (nx, B)))

    def configure(o):
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        o.set_fdyn(fdyn)
        o.set_cone_constraints(*cones)
        o.set_x_ref(xr)
        o.set_u_ref(ur)
        return o
    bs = configure(t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B))
    bs.set_warm_start(True)
    bs.set_precision(2)
    orcs = [configure(oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)) for _ in range(B)]
    x = x0.copy()
    for k in range(2):
        bs.set_x0(x)
        bs.solve()
        sol, st = bs.get_solution(), bs.get_status()
        xn = np.zeros_like(x)
        for b in range(B):
            o = orcs[b]
            o.set_x0(x[:, b])
            o.solve()
            r = o.get_solution()
            assert st["iter"][b] == r["iter"], f"solve {k} instance {b}"
            ex_, eu_ = nrel(sol["states"][:, :, b], r["x"]), nrel(sol["controls"][:, :, b], r["u"])
            assert ex_ <= FP32_TOL / 2 and eu_ <= FP32_TOL / 2, f"solve {k} instance {b}: x {ex_:.3e} u {eu_:.3e}"
            xn[:, b] = _f32(prob.A @ x[:, b] + prob.B @ r["u"][:, 0] + fdyn)
        x = xn
    for o in orcs:
        o.close()
    bs.close()


@pytest.mark.parametrize("seed", [1109, 1125, 1003, 1017])
def test_fuzz_seeds_that_missed_hold_1e5(hip_lib, oracle_built, seed):
    """scripts/fuzz_mfmat.py's random (6,3) families: seed 1109 (workspace kept: 4.2e-5 on mfmat, 1.4e-4 on the stream kernel) and
    1125 (closed loop: 5.1e-5) missed the sweep's 2e-5 / 3e-5 in round 3; under precision 2 they — and two seeds that passed —
    hold the suite's plain 1e-5 on solution, workspace and closed-loop steps"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_mfmat
    name, pattern, ok = fuzz_mfmat.one(seed, precision=2, tol=FP32_TOL)
    assert name == "generic<f64>" and ok, (seed, pattern)


def test_precision2_scope(hip_lib):
    """what precision 2 does not take is refused loudly: the fused closed loop, per-instance families"""
    prob = t.problems.cartpole(20, u_bound=0.5)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=64)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_precision(2)
    bs.set_x0(t.problems.cartpole_x0(64, seed=1))
    bs.solve()
    assert bs.kernel_name == "generic<f64>"
    with pytest.raises(t.TinyMPCError):
        bs.mpc_rollout(3)
    with pytest.raises(t.TinyMPCError):
        bs.set_precision(3)
    bs.close()
