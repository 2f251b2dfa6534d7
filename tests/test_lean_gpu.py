"""The lean kernel (csrc/admm_lean.hip.h, "lean<4,1,N>") — what BASELINE config 2 (the headline) runs on since round 4:
one-shot solves (cold start, workspace not kept) of the one-lane-per-instance cartpole entries with zero or shared
references and fp64 recurrences; the benchmark's pattern (no active state bound, zero references) is its leanest variant.
 * every instance against the fp64 oracle, fixed-iteration and tolerance-terminated (per-instance exits, check intervals),
   iteration counts / status / residuals included;
 * against the quad kernel it replaces for this calling pattern (TINYMPC_HIP_NO_LEAN keeps a solver there);
 * per-knot input bounds (the bounds then come from LDS knot by knot);
 * finite state bounds (state dual and q~ = vnew - g carried in fp32) and shared references (reference terms from LDS),
   alone and together, fixed-iteration and tolerance-terminated;
 * every calling pattern outside its scope stays on the quad kernel: per-instance references, warm starts / kept
   workspace, fp32 recurrences, adaptive rho, a cache whose AmBKt is not (A - B Kinf)'.
Reference arithmetic: src/codegen_src/tinympc/admm.cpp:13-107, solve() :109-207."""
import os

import numpy as np
import pytest

import tinympc_julia_amd as t
from tests.util import FP32_TOL, nrel, nrel_batch, parity_every_instance

pytestmark = pytest.mark.gpu

B_G1 = 24576          # one lane per instance from 20 480 instances up (select_quad_kernel)


def _oracle_make(oracle_built, prob, kw):
    def make(b=None):
        o = oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        return o
    return make


def _solver(prob, B, kw, warm=False):
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_warm_start(warm)
    return bs


@pytest.mark.parametrize("kw", [
    dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1),      # the benchmark's setting
    dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=47, check_termination=10),      # last check at 40, seven more iterations
    dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=1, check_termination=1),
    dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=30, check_termination=0),     # never check (the reference divides by it)
    dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1),    # per-instance exits
    dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=95, check_termination=10),
    dict(abs_pri_tol=1e-4, abs_dua_tol=1e-4, max_iter=100, check_termination=3),
], ids=["fixed100", "fixed47_ct10", "one_iter", "nocheck", "tol_ct1", "tol_ct10", "tol1e-4_ct3"])
def test_lean_vs_oracle_every_instance(hip_lib, oracle_built, kw):
    prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B_G1, seed=11)
    ref = oracle_built.solve_batch("orc64", prob, x0, nthreads=len(os.sched_getaffinity(0)), **kw)
    bs = _solver(prob, B_G1, kw)
    bs.set_x0(x0)
    status = bs.solve()
    assert bs.kernel_name == "quad<4,1,20,g1>" and bs.last_launch_name == "lean<4,1,20>"
    sol, st = bs.get_solution(), bs.get_status()
    assert status == int(np.any(st["solved"] == 0))
    same = parity_every_instance(sol, st, ref, _oracle_make(oracle_built, prob, kw), x0, kw, prob.rho, min_same=0.97, tag="lean")
    eq = st["iter"] == ref["iter"]
    # residuals as the reference reports them (types.hpp:128-131): the last check's values
    dres = np.abs(st["residuals"][eq] - ref["res"][eq]).max(axis=0) / np.maximum(1.0, np.abs(ref["res"][eq]).max(axis=0))
    assert dres.max() <= FP32_TOL, f"residuals (pri_x, dua_x, pri_u, dua_u) off by {dres}"
    if kw["abs_pri_tol"] > 0 and kw["check_termination"] > 0:
        assert len(np.unique(st["iter"])) > 3 and same >= 0.97
    else:
        assert np.all(st["iter"] == kw["max_iter"]) and not st["solved"].any()
    bs.close()


def test_lean_against_the_quad_kernel(hip_lib, monkeypatch):
    """the kernel this calling pattern ran on until round 3: same iterates to fp32 rounding of the stored state, same
    iteration counts wherever the residuals are not within rounding of the tolerance"""
    prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B_G1, seed=12)
    out = {}
    for which in ("lean", "quad"):
        if which == "quad":
            monkeypatch.setenv("TINYMPC_HIP_NO_LEAN", "1")
        for tag, kw in (("fixed", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)),
                        ("tol", dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1))):
            bs = _solver(prob, B_G1, kw)
            bs.set_x0(x0)
            bs.solve()
            assert bs.last_launch_name == ("lean<4,1,20>" if which == "lean" else "quad<4,1,20,g1>")
            out[which, tag] = (bs.get_solution(), bs.get_status())
            bs.close()
    monkeypatch.delenv("TINYMPC_HIP_NO_LEAN")
    for tag in ("fixed", "tol"):
        (sl, tl), (sq, tq) = out["lean", tag], out["quad", tag]
        eq = tl["iter"] == tq["iter"]
        assert eq.mean() >= (1.0 if tag == "fixed" else 0.995)
        ex = nrel_batch(sl["states"][:, :, eq], sq["states"][:, :, eq])
        eu = nrel_batch(sl["controls"][:, :, eq], sq["controls"][:, :, eq])
        assert ex.max() <= 4e-6 and eu.max() <= 4e-6, (tag, ex.max(), eu.max())
        assert np.array_equal(tl["solved"][eq], tq["solved"][eq])


def test_lean_per_knot_input_bounds(hip_lib, oracle_built):
    """u_min / u_max differ from knot to knot (set_bound_constraints takes nu x (N-1) matrices, bindings.cpp:378-411)"""
    prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B_G1, seed=13)
    rng = np.random.default_rng(5)
    prob.u_max = np.asfortranarray(0.2 + 0.5 * rng.random((1, 19)))
    prob.u_min = np.asfortranarray(-(0.2 + 0.5 * rng.random((1, 19))))
    for kw in (dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=1),
               dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=2)):
        ref = oracle_built.solve_batch("orc64", prob, x0, nthreads=len(os.sched_getaffinity(0)), **kw)
        bs = _solver(prob, B_G1, kw)
        bs.set_x0(x0)
        bs.solve()
        assert bs.last_launch_name == "lean<4,1,20>"
        sol, st = bs.get_solution(), bs.get_status()
        parity_every_instance(sol, st, ref, _oracle_make(oracle_built, prob, kw), x0, kw, prob.rho, min_same=0.97, tag="lean knot bounds")
        assert (sol["controls"] <= prob.u_max[:, :, None] + 1e-6).all() and (sol["controls"] >= prob.u_min[:, :, None] - 1e-6).all()
        bs.close()


def test_lean_unbounded_inputs_and_ragged_batch(hip_lib, oracle_built):
    """input bounds disabled (the clamp sees -inf / +inf); a batch that does not fill its last workgroup"""
    B = 20480 + 77
    prob, x0 = t.problems.cartpole(20), t.problems.cartpole_x0(B, seed=14)
    kw = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=25, check_termination=1)
    ref = oracle_built.solve_batch("orc64", prob, x0, nthreads=len(os.sched_getaffinity(0)), **kw)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_warm_start(False)
    bs.set_x0(x0)
    bs.solve()
    assert bs.last_launch_name == "lean<4,1,20>"
    sol = bs.get_solution()
    assert nrel_batch(sol["states"], ref["x"]).max() <= FP32_TOL and nrel_batch(sol["controls"], ref["u"]).max() <= FP32_TOL
    bs.close()


def test_lean_scope(hip_lib):
    """what the lean kernel does not take stays on the quad kernel (and a solver goes back and forth between the two)"""
    prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B_G1, seed=15)
    kw = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=10, check_termination=1)
    bs = _solver(prob, B_G1, kw)
    bs.set_x0(x0)
    bs.solve()
    assert bs.last_launch_name == "lean<4,1,20>"
    base = bs.get_solution()
    # kept workspace (the reference's default calling pattern)
    bs.set_warm_start(True)
    bs.reset()
    bs.solve()
    assert bs.last_launch_name == "quad<4,1,20,g1>"
    warm = bs.get_solution()
    assert nrel_batch(warm["controls"], base["controls"]).max() <= 4e-6
    bs.set_warm_start(False)
    # per-instance references
    xr = np.zeros((4, 20, B_G1), order="F"); xr[0] = 0.1
    bs.set_x_ref(xr)
    bs.solve()
    assert bs.last_launch_name == "quad<4,1,20,g1>"
    bs.set_x_ref(np.zeros((4, 20), order="F"))
    bs.solve()
    assert bs.last_launch_name == "lean<4,1,20>"
    # fp32 recurrences
    bs.set_precision(1)
    bs.solve()
    assert bs.last_launch_name == "quad<4,1,20,g1>"
    bs.set_precision(0)
    bs.solve()
    assert bs.last_launch_name == "lean<4,1,20>"
    again = bs.get_solution()
    assert np.array_equal(again["controls"], base["controls"]) and np.array_equal(again["states"], base["states"])
    # a cache whose AmBKt is no longer (A - B Kinf)': the two sweeps cannot share one matrix
    c = bs.get_cache_terms()
    bs.set_cache_terms(c["Kinf"], c["Pinf"], c["Quu_inv"], c["AmBKt"] * (1.0 + 1e-6))
    bs.solve()
    assert bs.last_launch_name == "quad<4,1,20,g1>"
    bs.set_cache_terms(c["Kinf"], c["Pinf"], c["Quu_inv"], c["AmBKt"])
    bs.solve()
    assert bs.last_launch_name == "lean<4,1,20>"
    bs.close()
    # small batches use four lanes per instance: no lean variant there
    b4 = _solver(prob, 512, kw)
    b4.set_x0(x0[:, :512])
    b4.solve()
    assert b4.last_launch_name == "quad<4,1,20,g4>"
    b4.close()


@pytest.mark.parametrize("N", [5, 10, 15])
def test_lean_other_horizons(hip_lib, oracle_built, N):
    """the other cartpole horizons with a one-lane-per-instance entry (tests/test_codegen.jl:15 uses N = 5, test_basic.jl N = 10)"""
    prob, x0 = t.problems.cartpole(N, u_bound=0.5), t.problems.cartpole_x0(B_G1, seed=20 + N)
    for kw in (dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1),
               dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)):
        ref = oracle_built.solve_batch("orc64", prob, x0, nthreads=len(os.sched_getaffinity(0)), **kw)
        bs = _solver(prob, B_G1, kw)
        bs.set_x0(x0)
        bs.solve()
        assert bs.kernel_name == f"quad<4,1,{N},g1>" and bs.last_launch_name == f"lean<4,1,{N}>"
        sol, st = bs.get_solution(), bs.get_status()
        parity_every_instance(sol, st, ref, _oracle_make(oracle_built, prob, kw), x0, kw, prob.rho, min_same=0.97, tag=f"lean N={N}")
        bs.close()


def test_routing_is_cached_between_solves(hip_lib):
    """the kernel selection is re-evaluated only when something it reads has changed (no getenv, no table walk per solve):
    a changed option re-routes at the next solve, an unchanged one keeps the cached decision"""
    prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B_G1, seed=16)
    bs = _solver(prob, B_G1, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=5, check_termination=1))
    bs.set_x0(x0)
    for _ in range(3):
        bs.solve()
        assert bs.kernel_name == "quad<4,1,20,g1>"
    bs.set_adaptive_rho(True)
    bs.solve()
    assert bs.kernel_name == "quad<4,1,20,g1>" and bs.last_launch_name == "quad<4,1,20,g1>"     # its adaptive variant
    bs.set_adaptive_rho(False)
    bs.set_fdyn(np.array([0.0, 0.01, 0.0, 0.0]))
    bs.solve()
    assert bs.kernel_name == "stream4<4,1>"
    bs.set_fdyn(np.zeros(4))
    bs.solve()
    assert bs.last_launch_name == "lean<4,1,20>"
    bs.close()


@pytest.mark.parametrize("case", ["state_bounds", "shared_refs", "state_bounds+shared_refs", "knot_state_bounds+refs"])
@pytest.mark.parametrize("setting", ["fixed", "tol"])
def test_lean_state_bounds_and_shared_references(hip_lib, oracle_built, case, setting):
    """finite state bounds (cartpole_example_reference_constrained.jl:16-18 bounds x_1) and shared references (examples:
    set_x_ref / set_u_ref) on the lean kernel: every instance against the oracle, bounds respected by the returned slack"""
    N = 20
    prob, x0 = t.problems.cartpole(N, u_bound=0.5), t.problems.cartpole_x0(B_G1, seed=31)
    rng = np.random.default_rng(7)
    xr = ur = None
    if "state_bounds" in case:
        prob.x_min, prob.x_max = prob.x_min.copy(), prob.x_max.copy()
        prob.x_max[0, :], prob.x_min[0, :] = 0.35, -0.35                  # binds: x0[0] is drawn from +-0.5
        prob.x_max[2, :], prob.x_min[2, :] = 0.08, -0.08
        if case.startswith("knot"):
            prob.x_max[0, N // 2:] = 0.2                                   # per-knot state bounds
            prob.u_max = prob.u_max.copy(); prob.u_max[:, ::3] = 0.35      # and per-knot input bounds
    if "refs" in case:
        xr = np.asfortranarray(0.1 * rng.standard_normal((4, N)))
        ur = np.asfortranarray(0.05 * rng.standard_normal((1, N - 1)))
    kw = (dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1) if setting == "fixed" else
          dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=2))
    ref = oracle_built.solve_batch("orc64", prob, x0, xref=xr, uref=ur, nthreads=len(os.sched_getaffinity(0)), **kw)

    def make(b=None):
        o = _oracle_make(oracle_built, prob, kw)()
        if xr is not None:
            o.set_x_ref(xr)
            o.set_u_ref(ur)
        return o
    bs = _solver(prob, B_G1, kw)
    if xr is not None:
        bs.set_x_ref(xr)
        bs.set_u_ref(ur)
    bs.set_x0(x0)
    bs.solve()
    assert bs.last_launch_name == "lean<4,1,20>"
    sol, st = bs.get_solution(), bs.get_status()
    parity_every_instance(sol, st, ref, make, x0, kw, prob.rho, min_same=0.97, tag=f"lean {case} {setting}")
    eq = (st["iter"] == ref["iter"]) & (st["solved"] == ref["solved"])
    dres = np.abs(st["residuals"][eq] - ref["res"][eq]).max(axis=0) / np.maximum(1.0, np.abs(ref["res"][eq]).max(axis=0))
    assert dres.max() <= 2 * FP32_TOL, f"residuals off by {dres}"
    if "state_bounds" in case:
        # (the kernel clamps to the fp32 value of a bound: float(0.2) is 3e-9 above 0.2)
        assert (sol["states"] <= prob.x_max[:, :, None] + 1e-7).all() and (sol["states"] >= prob.x_min[:, :, None] - 1e-7).all()
        assert np.abs(sol["states"][0]).max() >= 0.3499                    # the bound binds somewhere
    bs.close()


@pytest.mark.parametrize("case", ["plain", "state_bounds+shared_refs", "tol"])
def test_lean_beyond_one_wavefront_per_simd(hip_lib, oracle_built, case):
    """a batch of more than 256 workgroups x CUs instances: fixed-iteration solves take the 256-register kernels (two
    wavefronts per SIMD, the feed-forward term in fp32), tolerance-terminated ones the 512-register kernels in turn"""
    B, N = 70000, 20
    prob, x0 = t.problems.cartpole(N, u_bound=0.5), t.problems.cartpole_x0(B, seed=41)
    xr = ur = None
    if "state_bounds" in case:
        prob.x_min, prob.x_max = prob.x_min.copy(), prob.x_max.copy()
        prob.x_max[0, :], prob.x_min[0, :] = 0.35, -0.35
        rng = np.random.default_rng(9)
        xr, ur = np.asfortranarray(0.1 * rng.standard_normal((4, N))), np.asfortranarray(0.05 * rng.standard_normal((1, N - 1)))
    kw = (dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1) if case == "tol" else
          dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1))
    ref = oracle_built.solve_batch("orc64", prob, x0, xref=xr, uref=ur, nthreads=len(os.sched_getaffinity(0)), **kw)

    def make(b=None):
        o = _oracle_make(oracle_built, prob, kw)()
        if xr is not None:
            o.set_x_ref(xr)
            o.set_u_ref(ur)
        return o
    bs = _solver(prob, B, kw)
    if xr is not None:
        bs.set_x_ref(xr)
        bs.set_u_ref(ur)
    bs.set_x0(x0)
    bs.solve()
    assert bs.last_launch_name == "lean<4,1,20>"
    parity_every_instance(bs.get_solution(), bs.get_status(), ref, make, x0, kw, prob.rho, min_same=0.97, tag=f"lean 70 000 {case}")
    bs.close()


@pytest.mark.parametrize("state_bound", [False, True])
def test_kept_workspace_one_lane_per_instance(hip_lib, oracle_built, monkeypatch, state_bound):
    """the reference's default calling pattern (the workspace persists, admm.cpp:111-115) on the one-lane-per-instance quad
    kernel, whose workspace crosses the wavefront's LDS staging in both directions (admm_quad.hip.h: load_wave_x / _u,
    store_wave_x / _u): a ragged batch (the last wavefront has 33 instances), three warm-started solves of a host-stepped
    closed loop, tolerance-terminated — against persistent oracles on a sample that includes the ragged wavefront, the
    workspace arrays themselves included, and against the four-lanes-per-instance variant (plain strided loads / stores) on
    every instance"""
    N, B = 20, 20480 + 33
    prob = t.problems.cartpole(N, u_bound=0.5)
    if state_bound:
        prob.x_min, prob.x_max = prob.x_min.copy(), prob.x_max.copy()
        prob.x_min[0, :], prob.x_max[0, :] = -0.3, 0.3
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    x0 = t.problems.cartpole_x0(B, seed=21)
    pick = np.r_[0:16, 4090:4100, 20480:20480 + 33]
    runs = {}
    for which in ("g1", "g4"):
        if which == "g4":
            monkeypatch.setenv("TINYMPC_HIP_GROUP", "4")
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(**kw)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        x, out = x0.copy(), []
        for k in range(3):
            bs.set_x0(x)
            bs.solve()
            assert bs.last_launch_name == f"quad<4,1,20,{which}>", bs.last_launch_name
            sol, st, ws = bs.get_solution(), bs.get_status(), bs.get_workspace()
            out.append((sol, st, ws, x.copy()))
            x = np.asfortranarray(prob.A @ x + prob.B @ sol["controls"][:, 0, :])
        runs[which] = out
        bs.close()
    for k in range(3):
        (sa, ta, wa, _), (sb, tb, wb, _) = runs["g1"][k], runs["g4"][k]
        same = ta["iter"] == tb["iter"]
        assert same.mean() >= 0.999
        assert nrel_batch(sa["states"], sb["states"])[same].max() <= FP32_TOL and nrel_batch(sa["controls"], sb["controls"])[same].max() <= FP32_TOL
        for key in ("d", "y", "z", "g", "v"):
            scale = max(np.abs(wb[key]).max(), 1e-2)
            assert np.abs(wa[key] - wb[key])[:, :, same].max() <= 2e-5 * scale, (k, key)
    for b in pick:
        o = oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        for k in range(3):
            sol, st, ws, xk = runs["g1"][k]
            o.set_x0(xk[:, b])
            o.solve()
            r, sv = o.get_solution(), o.get_state()
            if r["iter"] != int(st["iter"][b]):
                break                                       # (a residual within rounding of the tolerance: the workspaces part ways)
            assert nrel(sol["states"][:, :, b], r["x"]) <= FP32_TOL and nrel(sol["controls"][:, :, b], r["u"]) <= FP32_TOL, (b, k)
            for key in ("d", "y", "z", "g", "v"):
                scale = max(np.abs(sv[key]).max(), 1e-2)
                lim = 4e-5 if key in ("g", "y") else 2e-5      # (the duals integrate the trajectory's per-iteration fp32 rounding: tests/test_mfmat_gpu.py)
                assert np.abs(ws[key][:, :, b] - sv[key]).max() <= lim * scale, (b, k, key)
        o.close()
