"""The reference's general constraint API on the transposed-sets matrix-core kernel: cone LISTS (bindings.cpp:453-490) and
linear-inequality rows (bindings.cpp:414-450).  The library's built-in `mfmat` entries compile one cone per side and no
rows; any other layout — two cones on a side, rows, a cone at other rows, a horizon the library was not built with — is
specialised at setup (csrc/jit.cpp: ONE kernel for exactly the solver's layout, compiled by hipcc as a child process,
cached) instead of falling to the HBM-streaming kernel.  Against the fp64 oracle, every instance by solution
(tests/util.parity_every_instance): the nine layouts of tests/test_mfmac_gpu.py::test_mfmac_general_cones_and_bounds
re-pointed at this path, then the calling patterns the built-in entries are tested with (workspace kept between solves,
the fused closed loop) on a layout with two cones and rows.  Cones / rows are the UNPINNED extensions (no reference
source): the oracle itself is pinned for them by tests/test_independent_optimum.py and tests/test_extensions_cpu.py."""
import os

import numpy as np
import pytest

import tinympc_julia_amd as t
from tests.test_mfmac_gpu import _loop, _oracle
from tests.util import FP32_TOL, nrel, parity_every_instance

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def jit_on(monkeypatch):
    monkeypatch.delenv("TINYMPC_HIP_NO_JIT", raising=False)
    cache = os.environ.get("TINYMPC_TEST_JIT_CACHE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "jit_cache")
    os.makedirs(cache, exist_ok=True)
    monkeypatch.setenv("TINYMPC_HIP_CACHE", os.path.abspath(cache))


def _family(seed, nx, nu, N):
    rng = np.random.default_rng(seed)
    A = np.eye(nx) + 0.15 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= 0.97 / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)),
                              np.diag(rng.uniform(0.5, 3.0, nu)), float(rng.uniform(0.5, 2.0)), N)
    prob.x_min, prob.x_max = -rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N)), rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N))
    prob.u_min, prob.u_max = -rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1)), rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1))
    return prob, rng


def _solver(prob, B, kw, xr, ur, fdyn, cones, lin, warm=False):
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if fdyn is not None:
        bs.set_fdyn(fdyn)
    if cones is not None:
        bs.set_cone_constraints(*cones)
    if lin is not None:
        bs.set_linear_constraints(*lin)
    bs.set_warm_start(warm)
    if xr is not None:
        bs.set_x_ref(xr)
        bs.set_u_ref(ur)
    return bs


CASES = {"cones_across_groups": 3, "state_cone_knot_bounds": 4, "two_state_cones_knot_bounds": 6, "two_state_cones_with_gap_and_input_cone": 7,
         "two_input_cones": 8, "linear_state_rows": 9, "linear_input_rows_only": 10, "cones_and_linear_both_sides": 11,
         "two_cones_each_side_and_rows": 12}


def _layout(case, prob, rng, nx, nu, N):
    cones, lin = None, None
    if case == "cones_across_groups":
        cones = ([1], [2], [0.8], [2], [4], [0.9])          # input rows 1..2; state rows 2..5
    elif case == "state_cone_knot_bounds":
        cones = ([], [], [], [3], [3], [1.2])               # a cone on the state side only, rows 3..5
        prob.x_min[:, N // 2:] -= 0.3                        # per-knot bounds
        prob.u_max[:, ::2] += 0.1
    elif case == "two_state_cones_knot_bounds":
        cones = ([], [], [], [0, 3], [3, 3], [1.1, 0.7])    # state rows 0..2 and 3..5
        prob.x_min[:, N // 2:] -= 0.3
        prob.u_max[:, ::2] += 0.1
    elif case == "two_state_cones_with_gap_and_input_cone":
        cones = ([0], [3], [0.6], [0, 4], [2, 2], [0.9, 1.3])   # state rows 0..1 and 4..5, rows 2..3 in no cone; input rows 0..2
    elif case == "two_input_cones":
        cones = ([0, 2], [2, 2], [0.8, 1.2], [1], [4], [0.9])   # input rows 0..1 and 2..3 (nu = 4); state rows 1..4
    elif case == "linear_state_rows":
        lin = (rng.standard_normal((2, nx)), [0.3, 0.5], np.zeros((0, nu)), [])
    elif case == "linear_input_rows_only":
        lin = (np.zeros((0, nx)), [], rng.standard_normal((3, nu)), [0.1, 0.2, 0.15])
    elif case == "cones_and_linear_both_sides":
        cones = ([0], [3], [0.6], [3], [3], [1.2])
        lin = (rng.standard_normal((1, nx)), [0.4], rng.standard_normal((2, nu)), [0.2, 0.1])
        prob.x_min[:, N // 2:] -= 0.3
    elif case == "two_cones_each_side_and_rows":
        cones = ([0, 2], [2, 2], [0.7, 1.1], [0, 3], [3, 3], [1.0, 0.8])
        lin = (rng.standard_normal((1, nx)), [0.35], rng.standard_normal((1, nu)), [0.15])
    return cones, lin


def _rows_bind(lin, ref, B):
    Ax, bx, Au, bu = lin
    X, U = ref["x"], ref["u"]
    act = sum(int((np.asarray(Ax) @ X[:, :, b] > np.asarray(bx)[:, None] - 1e-3).any()) for b in range(B)) if len(bx) else 0
    act += sum(int((np.asarray(Au) @ U[:, :, b] > np.asarray(bu)[:, None] - 1e-3).any()) for b in range(B)) if len(bu) else 0
    return act


@pytest.mark.parametrize("case", list(CASES))
def test_general_layouts_one_shot_vs_oracle(hip_lib, oracle_built, case):
    nx, nu, N, B = 6, (4 if case in ("two_input_cones", "two_cones_each_side_and_rows") else 3), 17, 29
    prob, rng = _family(CASES[case], nx, nu, N)
    fdyn = 0.02 * rng.standard_normal(nx)
    xr, ur = 0.2 * rng.standard_normal((nx, N)), 0.1 * rng.standard_normal((nu, N - 1))
    cones, lin = _layout(case, prob, rng, nx, nu, N)
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=50, check_termination=1)
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    mk = _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones, lin)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, fdyn, cones, lin)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name.startswith(f"mfmat<6,{nu},{N}>"), bs.kernel_name
    sol, st = bs.get_solution(), bs.get_status()
    parity_every_instance(sol, st, ref, mk, x0, kw, prob.rho, tag=case)
    if lin is not None:
        assert _rows_bind(lin, ref, B) > 0                  # the rows do bind somewhere (else the case tests nothing)
    # fixed iteration count, one check at the end (the other code path of the sets phase)
    kw2 = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=40, check_termination=10)
    mk2 = _oracle(oracle_built, prob, kw2, xr, ur, fdyn, cones, lin)
    ref2 = _loop(mk2, x0)
    bs.update_settings(**kw2, en_state_bound=1, en_input_bound=1)   # (update_settings sends every field, TinyMPC.jl:181-211)
    bs.solve()
    assert bs.kernel_name.startswith(f"mfmat<6,{nu},{N}>")
    parity_every_instance(bs.get_solution(), bs.get_status(), ref2, mk2, x0, kw2, prob.rho, tag=case + " fixed")
    bs.close()


ROCKET_GENERAL = dict(cones=([0], [3], [0.25], [0, 3], [3, 2], [0.5, 2.0]),       # thrust cone; glide-slope cone + |vx| <= 2 vy
                      lin=(np.array([[0.0, 0.0, 0.0, 0.0, 0.0, -1.0]]), [5.5], np.array([[1.0, 1.0, 0.0]]), [6.0]))   # -vz <= 5.5; ux + uy <= 6


@pytest.mark.parametrize("N,setting,steps", [(14, "tol", 5), (14, "fixed_ct7", 3), (25, "tol", 3)])
def test_general_layout_workspace_persists(hip_lib, oracle_built, N, setting, steps):
    """the reference's default calling pattern (the workspace carries over, admm.cpp:111-115) on the rocket with two state
    cones, the thrust cone and a row on both sides: a host-stepped closed loop beside one persistent oracle per instance —
    solution and iteration count after every solve (a converged exit leaves the previous iteration's slack behind for the
    next solve's first check, admm.cpp:181-197: the parked third set is part of that)"""
    B = 40
    prob = t.problems.rocket(N)
    xr, ur = t.problems.rocket_refs(N)
    cones, lin = ROCKET_GENERAL["cones"], ROCKET_GENERAL["lin"]
    kw = (dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1) if setting == "tol" else
          dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=45, check_termination=7))
    bs = _solver(prob, B, kw, xr, ur, prob.fdyn, cones, lin, warm=True)
    orcs = [_oracle(oracle_built, prob, kw, xr, ur, prob.fdyn, cones, lin)() for _ in range(B)]
    x = t.problems.rocket_x0(B, seed=5)
    converged = marginal = bound = 0
    for k in range(steps):
        bs.set_x0(x)
        bs.solve()
        assert bs.kernel_name.startswith(f"mfmat<6,3,{N}>"), bs.kernel_name
        sol, st = bs.get_solution(), bs.get_status()
        xn = np.zeros_like(x)
        for b in range(B):
            o = orcs[b]
            o.set_x0(x[:, b])
            o.solve()
            r = o.get_solution()
            converged += r["solved"]
            bound += int((lin[0] @ r["x"] > np.asarray(lin[1])[:, None] - 1e-3).any() or (lin[2] @ r["u"] > np.asarray(lin[3])[:, None] - 1e-3).any())
            ex_, eu_ = nrel(sol["states"][:, :, b], r["x"]), nrel(sol["controls"][:, :, b], r["u"])
            if int(st["iter"][b]) == r["iter"]:
                assert ex_ <= FP32_TOL and eu_ <= FP32_TOL, f"step {k} instance {b}: x {ex_:.3e} u {eu_:.3e}"
            else:                                           # a residual within rounding of the tolerance: the two workspaces part
                marginal += 1                               # ways from here on (the loop ends with this step)
                assert abs(int(st["iter"][b]) - r["iter"]) <= 1
                assert ex_ <= 3e-3 and eu_ <= 3e-3, f"step {k} instance {b} (one iteration apart): x {ex_:.3e} u {eu_:.3e}"
            xn[:, b] = prob.A @ x[:, b] + prob.B @ r["u"][:, 0] + prob.fdyn
        x = xn
        if marginal:
            break
    assert bound > 0                                        # the rows do bind
    if setting == "tol":
        assert converged >= 10                              # the converged-exit path is exercised
    assert marginal <= 2
    for o in orcs:
        o.close()
    bs.close()


def test_general_layout_fused_closed_loop(hip_lib):
    """the fused closed loop (examples/rocket_landing_constraints.jl:97-134 in one launch) on a specialised layout against the
    same loop stepped from the host on the same kernel"""
    N, B, steps = 14, 50, 5
    prob = t.problems.rocket(N)
    xr, ur = t.problems.rocket_refs(N)
    cones, lin = ROCKET_GENERAL["cones"], ROCKET_GENERAL["lin"]
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=60, check_termination=1)
    x0 = t.problems.rocket_x0(B, seed=6)
    fused = _solver(prob, B, kw, xr, ur, prob.fdyn, cones, lin, warm=True)
    fused.set_x0(x0)
    out = fused.mpc_rollout(steps)
    assert fused.kernel_name.startswith(f"mfmat<6,3,{N}>"), fused.kernel_name
    host = _solver(prob, B, kw, xr, ur, prob.fdyn, cones, lin, warm=True)
    x = x0.copy()
    for k in range(steps):
        host.set_x0(x)
        host.solve()
        sol, st = host.get_solution(), host.get_status()
        u0 = sol["controls"][:, 0, :]
        same = np.abs(out["iter"][k]) == st["iter"]
        assert same.mean() >= 0.95
        # (the fused loop carries the plant state in fp64 on chip, the host-stepped one hands it over as the solver's fp32 x0)
        scale = max(1.0, np.abs(u0).max())
        np.testing.assert_allclose(out["u"][:, k, same], u0[:, same], rtol=0, atol=1e-5 * scale)
        x = np.asfortranarray(prob.A @ x + prob.B @ u0 + prob.fdyn[:, None])
        np.testing.assert_allclose(out["x"][:, k, same], x[:, same], rtol=0, atol=1e-5 * max(1.0, np.abs(x).max()))
    fused.close()
    host.close()


def test_other_horizon_of_a_built_in_shape(hip_lib, oracle_built):
    """the rocket with its own cones and affine term at N = 25: no built-in `mfmat` entry -> specialised, not the stream kernel"""
    N, B = 25, 45
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=8)
    xr, ur = t.problems.rocket_refs(N)
    cones = ([0], [3], [0.25], [0], [3], [0.5])
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    mk = _oracle(oracle_built, prob, kw, xr, ur, prob.fdyn, cones)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, prob.fdyn, cones, None)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name.startswith("mfmat<6,3,25>"), bs.kernel_name
    parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, tag="rocket N=25")
    bs.close()


def test_layouts_the_kernel_does_not_take(hip_lib, monkeypatch):
    """three cones on a side, overlapping cones, TINYMPC_HIP_NO_JIT: the stream kernel as before, no error"""
    nx, nu, N, B = 6, 3, 17, 20
    prob, rng = _family(23, nx, nu, N)
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=20, check_termination=1)
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    for cones in (([], [], [], [0, 2, 4], [2, 2, 2], [1.0, 1.0, 1.0]), ([], [], [], [0, 2], [3, 3], [1.0, 1.0]), ([], [], [], [3, 0], [3, 3], [1.0, 1.0])):
        bs = _solver(prob, B, kw, None, None, None, cones, None)
        bs.set_x0(x0)
        bs.solve()
        assert bs.kernel_name.startswith("stream"), (cones, bs.kernel_name)
        bs.close()
    monkeypatch.setenv("TINYMPC_HIP_NO_JIT", "1")
    bs = _solver(prob, B, kw, None, None, None, ([], [], [], [0, 3], [3, 3], [1.0, 1.0]), None)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name.startswith("stream"), bs.kernel_name
    bs.close()
    monkeypatch.delenv("TINYMPC_HIP_NO_JIT")
    monkeypatch.setenv("TINYMPC_HIP_HIPCC", "/nonexistent/hipcc")     # no compiler on the machine: the same fallback, no error
    bs = _solver(prob, B, kw, None, None, None, ([], [], [], [0, 2], [2, 3], [1.0, 1.0]), None)   # (a layout no other test compiles)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name.startswith("stream"), bs.kernel_name
    bs.close()


@pytest.mark.parametrize("kernel", ["specialised", "stream", "generic"])
def test_a_side_switched_off_keeps_the_other_sides_rows(hip_lib, oracle_built, monkeypatch, kernel):
    """rows on both sides, then the state side switched off (tinympc_enable_linear; update_settings' en_state_linear = false,
    TinyMPC.jl:98-99): the input side's rows are the ones that act — the device pack holds the enabled sides only (until round
    4 the kernels looked for the input block behind rows that were not counted)"""
    if kernel != "specialised":
        monkeypatch.setenv("TINYMPC_HIP_NO_JIT", "1")
    if kernel == "generic":
        monkeypatch.setenv("TINYMPC_HIP_NO_STREAM", "1")
        monkeypatch.setenv("TINYMPC_HIP_NO_MFMAC", "1")
    nx, nu, N, B = 6, 3, 17, 29
    prob, rng = _family(10, nx, nu, N)
    fdyn = 0.02 * rng.standard_normal(nx)
    xr, ur = 0.2 * rng.standard_normal((nx, N)), 0.1 * rng.standard_normal((nu, N - 1))
    Ax, bx = rng.standard_normal((2, nx)), [0.3, 0.5]
    Au, bu = rng.standard_normal((3, nu)), [0.1, 0.2, 0.15]
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=50, check_termination=1)
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    only_u = (np.zeros((0, nx)), [], Au, bu)
    mk = _oracle(oracle_built, prob, kw, xr, ur, fdyn, None, only_u)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, fdyn, None, (Ax, bx, Au, bu))
    bs.enable_linear(0, 1)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name.startswith({"specialised": "mfmat<6,3,17>", "stream": "stream", "generic": "generic"}[kernel]), bs.kernel_name
    parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, tag=kernel)
    assert _rows_bind(only_u, ref, B) > 0
    bs.close()


@pytest.mark.parametrize("shape", [(5, 2, 9), (4, 2, 11), (8, 2, 12), (7, 1, 10), (5, 3, 14), (8, 4, 8), (7, 4, 9), (8, 3, 10), (4, 4, 12)])
def test_other_shapes_on_the_specialised_kernel(hip_lib, oracle_built, shape):
    """the kernel template beyond the rocket's (6,3): fewer than eight rows or one input row (hand-over stores under lane masks),
    a full second state slot (nx = 8), three and four inputs on the VALU columns ((8,3), (7,4), (8,4): more per-lane-group
    constants than a wavefront has lanes — loaded in one pass until this test found it), four inputs in slot 1 — affine term, a state cone, an
    input cone where two rows exist, one state row; one-shot, tolerance-terminated, every instance against the oracle"""
    nx, nu, N = shape
    B = 45
    prob, rng = _family(40 + nx * 4 + nu, nx, nu, N)
    fdyn = 0.02 * rng.standard_normal(nx)
    xr, ur = 0.2 * rng.standard_normal((nx, N)), 0.1 * rng.standard_normal((nu, N - 1))
    cones = ([0], [2], [0.8], [1], [3], [0.9]) if nu >= 2 else ([], [], [], [1], [3], [0.9])
    lin = (rng.standard_normal((1, nx)), [0.4], np.zeros((0, nu)), [])
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60, check_termination=1)
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    mk = _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones, lin)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, fdyn, cones, lin)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name.startswith(f"mfmat<{nx},{nu},{N}>"), bs.kernel_name
    parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, tag=str(shape))
    bs.close()


def test_general_layout_through_the_dropin_api(hip_lib, oracle_built):
    """the reference-named process-global entry points a Julia host binds (setup with fdyn and a batch, set_bound_constraints,
    set_cone_constraints with a cone LIST, set_linear_constraints, set_x_ref / set_u_ref, set_x0, solve): the layout is
    specialised at the first solve_mpc and the results are the handle API's, bit for bit, and the oracle's"""
    N, B = 14, 70
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=12)
    xr, ur = t.problems.rocket_refs(N)
    cones, lin = ROCKET_GENERAL["cones"], ROCKET_GENERAL["lin"]
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    bs = _solver(prob, B, kw, xr, ur, prob.fdyn, cones, lin)
    bs.set_x0(x0)
    bs.solve()
    want_name, ref_sol, ref_st = bs.kernel_name, bs.get_solution(), bs.get_status()
    bs.close()
    s = t.TinyMPCSolver()
    t.setup(s, prob.A, prob.B, prob.fdyn, prob.Q, prob.R, prob.rho, 6, 3, N, batch=B, max_iter=100, abs_pri_tol=2e-3, abs_dua_tol=1e-3)
    t.set_bound_constraints(s, prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    t.set_cone_constraints(s, *cones)
    t.set_linear_constraints(s, *lin)
    t.set_warm_start(s, False)
    t.set_x_ref(s, xr)
    t.set_u_ref(s, ur)
    t.set_x0(s, x0)
    t.solve(s)
    assert t.kernel_name() == want_name and want_name.startswith("mfmat<6,3,14> cx0:3+3:2 cu0:3 lin1,1"), (t.kernel_name(), want_name)
    got, st = t.get_solution(s), t.get_status(s)
    assert np.array_equal(got["states"], ref_sol["states"]) and np.array_equal(got["controls"], ref_sol["controls"])
    assert np.array_equal(st["iter"], ref_st["iter"]) and np.array_equal(st["solved"], ref_st["solved"])
    t.cleanup()
    mk = _oracle(oracle_built, prob, kw, xr, ur, prob.fdyn, cones, lin)
    parity_every_instance(ref_sol, ref_st, _loop(mk, x0), mk, x0, kw, prob.rho, tag="drop-in")


def test_general_layout_full_size_properties(hip_lib, oracle_built):
    """32 768 rocket instances (config 4's batch) at N = 20 with two state cones, the thrust cone and a row on both sides, on the
    unit specialised for the layout — through properties that do not need the oracle at that size: two launches agree bit for
    bit; the returned slack respects the box bounds exactly; where an instance converged (primal residuals below the
    tolerance) its cones and rows hold to within that tolerance; every 128th instance against the oracle"""
    N, B = 20, 32768
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=2)
    xr, ur = t.problems.rocket_refs(N)
    cones, lin = ROCKET_GENERAL["cones"], ROCKET_GENERAL["lin"]
    tol = 2e-3
    kw = dict(abs_pri_tol=tol, abs_dua_tol=1e-3, max_iter=150, check_termination=1)
    bs = _solver(prob, B, kw, xr, ur, prob.fdyn, cones, lin)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name.startswith("mfmat<6,3,20> cx0:3+3:2 cu0:3 lin1,1"), bs.kernel_name
    sol, st = bs.get_solution(), bs.get_status()
    bs.solve()
    sol2, st2 = bs.get_solution(), bs.get_status()
    assert np.array_equal(sol["states"], sol2["states"]) and np.array_equal(sol["controls"], sol2["controls"]) and np.array_equal(st["iter"], st2["iter"])
    X, U = sol["states"], sol["controls"]
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    assert (X >= f32(prob.x_min)[:, :, None]).all() and (X <= f32(prob.x_max)[:, :, None]).all()
    assert (U >= f32(prob.u_min)[:, :, None]).all() and (U <= f32(prob.u_max)[:, :, None]).all()
    done = st["solved"] == 1
    assert done.sum() >= 0.2 * B                             # (the hard instances run into max_iter: the rows bind for them)
    Xc, Uc = X[:, :, done], U[:, :, done]
    # the box slack is within the primal tolerance of x, and x within it of every other set's slack (admm.cpp:93-96)
    slackx, slacku = 2 * tol * np.abs(X).max(), 2 * tol * np.abs(U).max()
    Acu, qcu, mu_u, Acx, qcx, mu_x = cones
    for a0, q, mu in zip(Acx, qcx, mu_x):
        head, axis = Xc[a0:a0 + q - 1], Xc[a0 + q - 1]
        assert (np.sqrt((head ** 2).sum(axis=0)) <= mu * axis + (1 + mu) * slackx).all()
    for a0, q, mu in zip(Acu, qcu, mu_u):
        head, axis = Uc[a0:a0 + q - 1], Uc[a0 + q - 1]
        assert (np.sqrt((head ** 2).sum(axis=0)) <= mu * axis + (1 + mu) * slacku).all()
    Ax, bx, Au, bu = lin
    assert (np.einsum("rk,knb->rnb", np.asarray(Ax), Xc) <= np.asarray(bx)[:, None, None] + np.abs(Ax).sum() * slackx).all()
    assert (np.einsum("rk,knb->rnb", np.asarray(Au), Uc) <= np.asarray(bu)[:, None, None] + np.abs(Au).sum() * slacku).all()
    pick = np.arange(0, B, 128)
    mk = _oracle(oracle_built, prob, kw, xr, ur, prob.fdyn, cones, lin)
    ref = _loop(mk, x0[:, pick])
    sub = dict(states=X[:, :, pick], controls=U[:, :, pick])
    sst = dict(iter=st["iter"][pick], solved=st["solved"][pick], residuals=st["residuals"][pick])
    parity_every_instance(sub, sst, ref, mk, x0[:, pick], kw, prob.rho, min_same=0.9, tag="full-size sample")
    bs.close()
