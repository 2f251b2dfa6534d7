"""The matrix-core kernels for one-shot solves with box bounds, the affine dynamics term and second-order cones —
BASELINE config 4's path: the LDS-resident one with a run-time horizon (csrc/admm_mfmac.hip.h, "mfmac<6,3>") and the
register-resident one with the horizon compiled in (csrc/admm_mfmar.hip.h, "mfmar<6,3,50>", what config 4 itself runs
on) — against the fp64 oracle, every instance by solution
(tests/util.parity_every_instance), and against the run-time-horizon stream kernel it replaces for these solves.
Cones / fdyn are the UNPINNED extensions (no reference source): the oracle itself is pinned for them by
tests/test_independent_optimum.py and tests/test_extensions_cpu.py."""
import numpy as np
import pytest

import tinympc_julia_amd as t
from tests.util import FP32_TOL, nrel_batch, parity_every_instance

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _three_wavefront_kernels_only(monkeypatch):
    """since round 3 these problems run on the transposed-sets kernel (tests/test_mfmat_gpu.py); this module keeps the
    three-wavefront kernels behind it covered"""
    monkeypatch.setenv("TINYMPC_HIP_NO_MFMAT", "1")


def _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones, lin=None):
    def make(b=None):
        o = oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if fdyn is not None:
            o.set_fdyn(fdyn)
        if cones is not None:
            o.set_cone_constraints(*cones)
        if lin is not None:
            o.set_linear_constraints(*lin)
        if xr is not None:
            o.set_x_ref(xr)
            o.set_u_ref(ur)
        return o
    return make


def _loop(make, x0):
    B = x0.shape[1]
    out = None
    for b in range(B):
        o = make(b)
        o.set_x0(x0[:, b])
        o.solve()
        r = o.get_solution()
        if out is None:
            out = dict(x=np.zeros(r["x"].shape + (B,)), u=np.zeros(r["u"].shape + (B,)), iter=np.zeros(B, dtype=int),
                       solved=np.zeros(B, dtype=int), res=np.zeros((B, 4)))
        out["x"][:, :, b], out["u"][:, :, b] = r["x"], r["u"]
        out["iter"][b], out["solved"][b], out["res"][b] = r["iter"], r["solved"], r["res"]
        o.close()
    return out


def _solver(prob, B, kw, xr, ur, fdyn, cones, lin=None):
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if fdyn is not None:
        bs.set_fdyn(fdyn)
    if cones is not None:
        bs.set_cone_constraints(*cones)
    if lin is not None:
        bs.set_linear_constraints(*lin)
    bs.set_warm_start(False)                               # one-shot: cold start, workspace not kept
    if xr is not None:
        bs.set_x_ref(xr)
        bs.set_u_ref(ur)
    return bs


ROCKET_CONES = ([0], [3], [0.25], [0], [3], [0.5])          # inputs first (bindings.cpp:453-459)
SETTINGS = {
    "fixed60": dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=1),
    "tol": dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1),   # rocket_landing_constraints.jl:61-62
    "tol_ct10": dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=95, check_termination=10),
    "fixed_ct7": dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=45, check_termination=7),   # last check at 42, three more iterations
    "nocheck": dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=30, check_termination=0),
}


def _kernel_for(N, lds_only=False):
    return f"mfmar<6,3,{N}>" if N in (10, 20, 30, 50) and not lds_only else "mfmac<6,3>"


@pytest.mark.parametrize("N", [10, -10, 50, -50, 20, 30, 23, 2])
@pytest.mark.parametrize("mode", ["fdyn+cones", "fdyn", "cones"])
@pytest.mark.parametrize("setting", list(SETTINGS))
def test_mfmac_rocket_vs_oracle(hip_lib, oracle_built, monkeypatch, N, mode, setting):
    if N in (23, 2, -50, 20, 30) and (mode != "fdyn+cones" or setting not in ("fixed60", "tol")):
        pytest.skip("the odd horizons (and the LDS kernel at N = 50) run the two main settings only")
    lds_only = N < 0                                        # -10, -50: the compiled horizons on the run-time-horizon kernel
    N = abs(N)
    if lds_only:
        monkeypatch.setenv("TINYMPC_HIP_NO_MFMAR", "1")
    B = 37                                                  # ragged: two full wavefronts of 16 and one of 5
    kw = SETTINGS[setting]
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=2)
    xr, ur = t.problems.rocket_refs(N)
    fdyn = prob.fdyn if "fdyn" in mode else None
    cones = ROCKET_CONES if "cones" in mode else None
    mk = _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, fdyn, cones)
    bs.set_x0(x0)
    status = bs.solve()
    assert bs.kernel_name == _kernel_for(N, lds_only)
    sol, st = bs.get_solution(), bs.get_status()
    assert status == int(np.any(st["solved"] == 0))
    parity_every_instance(sol, st, ref, mk, x0, kw, prob.rho, tag=f"N={N} {mode} {setting}")
    same = st["iter"] == ref["iter"]
    if kw["check_termination"] > 0:
        # residuals are differences of fp32 values of magnitude up to ~100 (thrust): a few ulp of those is the floor
        atol = 4e-7 * max(1.0, np.abs(ref["x"]).max(), np.abs(ref["u"]).max()) * max(1.0, prob.rho)
        assert np.allclose(st["residuals"][same], ref["res"][same], rtol=1e-2, atol=atol), \
            np.abs(st["residuals"][same] - ref["res"][same]).max()
    # a second solve of the same inputs returns the same bits (nothing of the first one survives)
    bs.solve()
    assert np.array_equal(bs.get_solution()["controls"], sol["controls"])
    assert np.array_equal(bs.get_status()["iter"], st["iter"])
    bs.close()


@pytest.mark.parametrize("setting", ["fixed60", "tol"])
def test_mfmac_equals_stream_kernel(hip_lib, monkeypatch, setting):
    """same solve on the HBM-streaming kernel it replaces: same arithmetic, different storage and mat-vec grouping"""
    N, B = 50, 70
    kw = SETTINGS[setting]
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=12)
    xr, ur = t.problems.rocket_refs(N)
    outs = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("TINYMPC_HIP_NO_MFMAC", env)
        bs = _solver(prob, B, kw, xr, ur, prob.fdyn, ROCKET_CONES)
        bs.set_x0(x0)
        bs.solve()
        outs.append((bs.kernel_name, bs.get_solution(), bs.get_status()))
        bs.close()
    assert outs[0][0] == "mfmar<6,3,50>" and outs[1][0] == "stream4<6,3>"
    same = outs[0][2]["iter"] == outs[1][2]["iter"]
    assert same.mean() >= 0.95
    assert nrel_batch(outs[0][1]["states"], outs[1][1]["states"])[same].max() <= 3e-6
    assert nrel_batch(outs[0][1]["controls"], outs[1][1]["controls"])[same].max() <= 3e-6


@pytest.mark.parametrize("case", ["cones_across_groups", "state_cone_knot_bounds", "zero_refs_box_only"])
def test_mfmac_general_cones_and_bounds(hip_lib, oracle_built, monkeypatch, case):
    """cones whose rows sit in different lane groups and slots (rows 2..5 = slot 0 of groups 2, 3 and slot 1 of groups
    0, 1), a cone on one side only, a 2-row input cone; bounds that depend on the knot; no references.  (Two cones on a
    side and linear-inequality rows — bindings.cpp:414-490 — ran here on request in round 3; since round 4 they run on the
    transposed-sets kernel specialised for the layout: tests/test_mfmat_general_gpu.py holds those six cases.)"""
    rng = np.random.default_rng({"cones_across_groups": 3, "state_cone_knot_bounds": 4, "zero_refs_box_only": 5}[case])
    nx, nu, N, B = 6, 3, 17, 29
    A = np.eye(nx) + 0.15 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= 0.97 / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)),
                              np.diag(rng.uniform(0.5, 3.0, nu)), float(rng.uniform(0.5, 2.0)), N)
    prob.x_min, prob.x_max = -rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N)), rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N))
    prob.u_min, prob.u_max = -rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1)), rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1))
    fdyn = 0.02 * rng.standard_normal(nx)
    xr, ur = 0.2 * rng.standard_normal((nx, N)), 0.1 * rng.standard_normal((nu, N - 1))
    cones = None
    if case == "cones_across_groups":
        cones = ([1], [2], [0.8], [2], [4], [0.9])          # input rows 1..2; state rows 2..5
    elif case == "state_cone_knot_bounds":
        cones = ([], [], [], [3], [3], [1.2])               # a cone on the state side only, rows 3..5
        prob.x_min[:, N // 2:] -= 0.3                        # per-knot bounds: the pack keeps every knot
        prob.u_max[:, ::2] += 0.1
    else:
        monkeypatch.setenv("TINYMPC_HIP_MFMAC_ALL", "1")    # box-only one-shot solve of a run-time-horizon shape
        fdyn, xr, ur = None, None, None
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=50, check_termination=1)
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    mk = _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, fdyn, cones)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name == "mfmac<6,3>"
    sol, st = bs.get_solution(), bs.get_status()
    parity_every_instance(sol, st, ref, mk, x0, kw, prob.rho, tag=case)
    bs.close()


def test_mfmar_box_only_other_horizons(hip_lib, oracle_built):
    """box-only one-shot solves at N = 20 (no quad instantiation: the stream kernel otherwise) run on mfmar and match the oracle"""
    N, B = 20, 33
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=5)
    xr, ur = t.problems.rocket_refs(N)
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60, check_termination=1)
    mk = _oracle(oracle_built, prob, kw, xr, ur, None, None)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, None, None)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name == "mfmar<6,3,20>"
    parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, tag="box only N=20")
    bs.close()


@pytest.mark.parametrize("case", ["knot_bounds", "zero_refs_input_cone", "state_cone_outside_slot0", "box_only"])
def test_mfmar_variants(hip_lib, oracle_built, case):
    """the compiled-horizon kernel's other instantiations at N = 50: bounds that depend on the knot (LDS pack), no
    references + a cone on the input side only, box sets only; a state cone outside rows 0..3 is not its case (-> the
    LDS kernel)"""
    N, B = 50, 21
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=8)
    xr, ur = t.problems.rocket_refs(N)
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=80, check_termination=1)
    cones, fdyn, expect = ROCKET_CONES, prob.fdyn, "mfmar<6,3,50>"
    if case == "knot_bounds":
        prob.x_min = prob.x_min.copy()
        prob.u_max = prob.u_max.copy()
        prob.x_min[:, N // 2:] -= 0.5
        prob.u_max[:, ::3] += 1.0
    elif case == "zero_refs_input_cone":
        xr, ur, cones = None, None, ([0], [3], [0.25], [], [], [])
    elif case == "box_only":
        cones, fdyn = None, None                            # plain one-shot solve of the compiled shape: mfmar, not the quad kernel
    else:
        cones = ([0], [3], [0.25], [2], [3], [0.6])         # state rows 2..4: slot 0 and slot 1
        expect = "mfmac<6,3>"
    mk = _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, fdyn, cones)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name == expect
    parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, tag=case)
    bs.close()


def test_mfmac_is_not_used_where_it_does_not_apply(hip_lib, monkeypatch):
    """warm-started / workspace-keeping solves and per-instance references stay on the stream kernel; so do two cones on a
    side and linear rows (with the specialisation at setup off, as in this suite: tests/conftest.py)"""
    prob = t.problems.rocket(20)
    xr, ur = t.problems.rocket_refs(20)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=8)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_fdyn(prob.fdyn)
    bs.set_cone_constraints(*ROCKET_CONES)
    bs.set_x0(t.problems.rocket_x0(8, seed=1))
    bs.solve()                                              # default: the workspace persists (reference semantics)
    assert bs.kernel_name == "stream4<6,3>"
    bs.set_warm_start(False)
    bs.solve()
    assert bs.kernel_name == "mfmar<6,3,20>"
    bs.set_x_ref(np.repeat(xr[:, :, None], 8, axis=2))      # per-instance references
    bs.set_u_ref(np.repeat(ur[:, :, None], 8, axis=2))
    bs.solve()
    assert bs.kernel_name == "stream4<6,3>"
    bs.set_x_ref(xr)
    bs.set_u_ref(ur)
    bs.set_linear_constraints(np.array([[0.0, 0.0, 0.0, 0.0, 0.0, -1.0]]), [2.5], np.zeros((0, 3)), [])
    bs.solve()
    assert bs.kernel_name == "stream4<6,3>"
    bs.close()
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=8)   # two cones on a side
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_cone_constraints([0], [3], [0.25], [0, 3], [3, 3], [0.5, 1.5])
    bs.set_warm_start(False)
    bs.set_x0(t.problems.rocket_x0(8, seed=1))
    bs.solve()
    assert bs.kernel_name == "stream4<6,3>"
    bs.close()
