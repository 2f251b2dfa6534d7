"""CPU tests of the oracle itself (test infrastructure pinned before it is trusted):
the fp64 C restatement must reproduce every golden vector produced by the compiled reference
snapshot to <= 1e-12 (norm-relative), the fp32+fp64-cache model must stay within the fp32 parity
tolerance, and where the reference library itself is present (dev container) it is re-run live.
"""
import numpy as np
import pytest

import tinympc_julia_amd as t
from tests.util import FP64_TOL, cm, golden_names, load_golden, nrel, problem_of

SINGLE = [n for n in golden_names() if n.startswith(("G1_", "G3", "G4"))]
BATCH = [n for n in golden_names() if n.startswith(("G2_", "G6", "G7_"))]
MPC = [n for n in golden_names() if n.startswith("G5")]
TRACE = [n for n in golden_names() if n.startswith("G8")]


def test_golden_inventory():
    """SURVEY.md §8(c) lists G1..G8; all are committed."""
    names = golden_names()
    for prefix in ("G1_", "G2_", "G3a", "G3b", "G3c", "G3d", "G4_", "G5_", "G6_", "G7_", "G8a", "G8b"):
        assert any(n.startswith(prefix) for n in names), prefix


def _mk(oracle, kind, g, prob=None):
    prob = prob or problem_of(g)
    s = oracle.CpuSolver(kind, prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
    s.update_settings(**g["settings"])
    if prob.has_bounds():
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    return s, prob


def _cmp(o, exp, prob, tol):
    assert nrel(o["x"], cm(exp["x"], prob.nx, prob.N)) <= tol
    assert nrel(o["u"], cm(exp["u"], prob.nu, prob.N - 1)) <= tol


@pytest.mark.parametrize("name", SINGLE)
def test_orc64_single(oracle_built, name):
    g = load_golden(name)
    s, prob = _mk(oracle_built, "orc64", g)
    if g["xref"] is not None:
        s.set_x_ref(cm(g["xref"], prob.nx, prob.N))
    if g["uref"] is not None:
        s.set_u_ref(cm(g["uref"], prob.nu, prob.N - 1))
    s.set_x0(g["x0"])
    status = s.solve()
    o = s.get_solution()
    exp = g["expect"]
    assert (status, o["iter"], o["solved"]) == (exp["status"], exp["iter"], exp["solved"])
    _cmp(o, exp, prob, FP64_TOL)
    assert np.allclose(o["res"], exp["res"], rtol=1e-9, atol=1e-13)
    c = s.get_cache()
    for key, (r, cc) in dict(Kinf=(prob.nu, prob.nx), Pinf=(prob.nx, prob.nx), Quu_inv=(prob.nu, prob.nu),
                             AmBKt=(prob.nx, prob.nx)).items():
        assert nrel(c[key], cm(g["cache"][key], r, cc)) <= FP64_TOL, key
    st = s.get_state()
    for key in ("d", "y", "g", "v", "z"):
        ref = np.asarray(g["state_after"][key])
        assert np.abs(st[key].flatten(order="F") - ref).max() <= FP64_TOL * max(1.0, np.abs(ref).max()), key


def test_known_answer_config1(oracle_built):
    """SURVEY.md §8(c) known-answer for examples/cartpole_example_one_solve.jl."""
    g = load_golden("G1_cartpole_one_solve")
    e = g["expect"]
    assert (e["status"], e["iter"]) == (0, 7)
    assert abs(e["u"][0] - 1.17807593) < 5e-9 and abs(e["u"][18] + 0.67943572) < 5e-9
    assert np.allclose(g["cache"]["Kinf"], [-1.8281816031, -2.4111848780, 20.6738188203, 3.3664150316], atol=5e-10)


@pytest.mark.parametrize("name", BATCH)
def test_orc64_batch(oracle_built, name):
    g = load_golden(name)
    prob = problem_of(g)
    B = g["batch"]
    xr = None if g["xref"] is None else cm(g["xref"], prob.nx, prob.N)
    ur = None if g["uref"] is None else cm(g["uref"], prob.nu, prob.N - 1)
    r = oracle_built.solve_batch("orc64", prob, cm(g["x0"], prob.nx, B), xref=xr, uref=ur, nthreads=2,
                                 **{k: g["settings"][k] for k in ("abs_pri_tol", "abs_dua_tol", "max_iter",
                                                                  "check_termination")})
    for b, exp in enumerate(g["expect"]):
        assert (int(r["iter"][b]), int(r["solved"][b])) == (exp["iter"], exp["solved"])
        _cmp(dict(x=r["x"][:, :, b], u=r["u"][:, :, b]), exp, prob, FP64_TOL)


@pytest.mark.parametrize("name", MPC)
def test_orc64_warm_start_sequence(oracle_built, name):
    """Workspace persistence across solves, closed loop x+ = A x + B u0."""
    g = load_golden(name)
    s, prob = _mk(oracle_built, "orc64", g)
    s.set_x0(g["x0"])
    s.set_x_ref(np.zeros((prob.nx, prob.N)))
    s.set_u_ref(np.zeros((prob.nu, prob.N - 1)))
    x = np.array(g["x0"], dtype=np.float64)
    for step in g["steps"]:
        assert np.abs(x - np.array(step["x0"])).max() <= 1e-12
        status = s.solve()
        o = s.get_solution()
        assert (status, o["iter"]) == (step["status"], step["iter"])
        _cmp(o, step, prob, FP64_TOL)
        st = s.get_state()
        for key in ("d", "y", "g", "v", "z"):
            ref = np.asarray(step["state_after"][key])
            assert np.abs(st[key].flatten(order="F") - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        x = prob.A @ x + prob.B @ o["u"][:, 0]
        s.set_x0(x)


@pytest.mark.parametrize("name", TRACE)
def test_orc64_residual_trace(oracle_built, name):
    g = load_golden(name)
    prob = problem_of(g)
    for tr in g["trace"][::7] + [g["trace"][-1]]:
        s = oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        s.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=tr["k"], check_termination=1)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x0(g["x0"])
        assert s.solve() == tr["status"]
        o = s.get_solution()
        assert o["iter"] == tr["iter"]
        assert np.allclose(o["res"], tr["res"], rtol=1e-9, atol=1e-13)
        assert np.abs(o["u"][:, 0] - np.array(tr["u0"])).max() <= 1e-12 * max(1.0, np.abs(tr["u0"]).max())


@pytest.mark.parametrize("name", ["G2_cartpole_box_fixed100", "G6_quadrotor_box_fixed100", "G7_rocket_box_fixed100"])
def test_orc32_model_within_fp32_budget(oracle_built, name):
    """fp32 loop + fp64-computed cache (the all-fp32 kernel's arithmetic model): a few 1e-6 on these
    fixtures — and the reason the kernel's default keeps the recurrences in fp64."""
    g = load_golden(name)
    prob = problem_of(g)
    B = g["batch"]
    xr = None if g["xref"] is None else cm(g["xref"], prob.nx, prob.N)
    ur = None if g["uref"] is None else cm(g["uref"], prob.nu, prob.N - 1)
    r = oracle_built.solve_batch("orc32", prob, cm(g["x0"], prob.nx, B), xref=xr, uref=ur,
                                 abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100)
    for b, exp in enumerate(g["expect"]):
        assert nrel(r["x"][:, :, b], cm(exp["x"], prob.nx, prob.N)) <= 2e-5
        assert nrel(r["u"][:, :, b], cm(exp["u"], prob.nu, prob.N - 1)) <= 2e-5


def test_live_reference_when_present(oracle_built):
    """Dev container only: the compiled reference snapshot (oracle/_ref) re-run against the restatement
    on fresh random inputs — guards the fixtures against staleness.  Skipped on the GPU box / CI where
    /root/reference (and so the prebuilt .so) may be absent."""
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(123)
    for prob, x0 in ((t.problems.cartpole(20, u_bound=0.4), t.problems.cartpole_x0(16, seed=77)),
                     (t.problems.quadrotor(20), t.problems.quadrotor_x0(8, seed=78))):
        kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60)
        a = oracle_built.solve_batch("ref", prob, x0, **kw)
        b = oracle_built.solve_batch("orc64", prob, x0, **kw)
        assert np.array_equal(a["iter"], b["iter"]) and np.array_equal(a["solved"], b["solved"])
        assert np.abs(a["x"] - b["x"]).max() <= 1e-11 and np.abs(a["u"] - b["u"]).max() <= 1e-11
    del rng
