"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.json from the compiled
reference snapshot (oracle/_ref/libtinympc_ref.so, built from /root/reference in
place by oracle/Makefile).  Run in the dev container:

    python -m oracle.make_golden

Fixtures are data only (inputs, settings, expected outputs in fp64); SURVEY.md
§8(c) lists the cases G1..G8.  The reference cannot travel to the GPU box, the
fixtures do.
"""
import importlib.util
import json
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)

from oracle.cpu_oracle import CpuSolver, build  # noqa: E402


def _problems():
    spec = importlib.util.spec_from_file_location(
        "_tmpc_problems", os.path.join(_ROOT, "tinympc-julia_amd", "problems.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["_tmpc_problems"] = m
    spec.loader.exec_module(m)
    return m


P = _problems()
OUT = os.path.join(_ROOT, "tests", "golden")


def _l(a):
    """column-major flatten -> list (JSON keeps full repr precision)"""
    return np.asarray(a, dtype=np.float64).flatten(order="F").tolist()


def _prob_dict(prob):
    d = dict(name=prob.name, nx=prob.nx, nu=prob.nu, N=prob.N, rho=prob.rho,
             A=_l(prob.A), B=_l(prob.B), Q=_l(prob.Q), R=_l(prob.R))
    if prob.has_bounds():
        d.update(x_min=_l(prob.x_min), x_max=_l(prob.x_max), u_min=_l(prob.u_min), u_max=_l(prob.u_max))
    return d


def _mk(prob, settings):
    s = CpuSolver("ref", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
    s.update_settings(**settings)
    if prob.has_bounds():
        # set_bound_constraints auto-enables both flags (bindings.cpp:400-404); it is called
        # after setup()'s update_settings in every reference script.
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    return s


def _sol(s, status):
    o = s.get_solution()
    return dict(status=status, iter=o["iter"], solved=o["solved"], x=_l(o["x"]), u=_l(o["u"]),
                res=_l(o["res"]))


def case_single(name, prob, x0, settings, xref=None, uref=None, note=""):
    s = _mk(prob, settings)
    if xref is not None:
        s.set_x_ref(xref)
    if uref is not None:
        s.set_u_ref(uref)
    s.set_x0(x0)
    st = s.solve()
    d = dict(case=name, note=note, problem=_prob_dict(prob), settings=settings, x0=_l(x0),
             xref=None if xref is None else _l(xref), uref=None if uref is None else _l(uref),
             expect=_sol(s, st), cache={k: _l(v) for k, v in s.get_cache().items()},
             state_after={k: _l(v) for k, v in s.get_state().items()})
    return d


def case_batch(name, prob, x0s, settings, xref=None, uref=None, note=""):
    """Several cold-start instances of one family; x0s (nx, B)."""
    inst = []
    cache = None
    for b in range(x0s.shape[1]):
        s = _mk(prob, settings)
        if xref is not None:
            s.set_x_ref(xref)
        if uref is not None:
            s.set_u_ref(uref)
        s.set_x0(x0s[:, b])
        st = s.solve()
        inst.append(_sol(s, st))
        cache = {k: _l(v) for k, v in s.get_cache().items()}
    return dict(case=name, note=note, problem=_prob_dict(prob), settings=settings, x0=_l(x0s),
                batch=int(x0s.shape[1]), xref=None if xref is None else _l(xref),
                uref=None if uref is None else _l(uref), expect=inst, cache=cache)


def case_mpc(name, prob, x0, settings, steps, note=""):
    """Warm-start closed loop: solve -> u0 -> x+ = A x + B u0 -> set_x0 (cartpole_example_mpc.jl:35-51)."""
    s = _mk(prob, settings)
    s.set_x0(x0)
    s.set_x_ref(np.zeros((prob.nx, prob.N)))
    s.set_u_ref(np.zeros((prob.nu, prob.N - 1)))
    x = np.array(x0, dtype=np.float64)
    seq = []
    for _ in range(steps):
        st = s.solve()
        o = _sol(s, st)
        o["x0"] = _l(x)
        o["state_after"] = {k: _l(v) for k, v in s.get_state().items()}
        seq.append(o)
        u0 = np.array(o["u"][: prob.nu])
        x = prob.A @ x + prob.B @ u0
        s.set_x0(x)
    return dict(case=name, note=note, problem=_prob_dict(prob), settings=settings, x0=_l(x0),
                steps=seq)


def case_trace(name, prob, x0, iters, note=""):
    """Per-iteration residual/solution trace: cold solve with max_iter = k, tol 0, k = 1..iters."""
    tr = []
    for k in range(1, iters + 1):
        st = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=k, check_termination=1)
        s = _mk(prob, st)
        s.set_x0(x0)
        code = s.solve()
        o = s.get_solution()
        tr.append(dict(k=k, status=code, iter=o["iter"], res=_l(o["res"]),
                       u0=_l(o["u"][:, 0]), xN=_l(o["x"][:, -1])))
    return dict(case=name, note=note, problem=_prob_dict(prob), x0=_l(x0), trace=tr)


def _fd_sensitivity(prob, h=1e-6):
    """Sensitivities handed to the reference as INPUT for the cartpole case: forward differences of the
    rho-regularised LQR gains, the recipe of TinyMPC.jl:301-352 in numpy."""
    def lqr(rho):
        nx, nu = prob.nx, prob.nu
        Qr, Rr = prob.Q + rho * np.eye(nx), prob.R + rho * np.eye(nu)
        Pm, K = Qr.copy(), np.zeros((nu, nx))
        for it in range(1, 5001):
            Kp = K
            K = np.linalg.solve(Rr + prob.B.T @ Pm @ prob.B + 1e-8 * np.eye(nu), prob.B.T @ Pm @ prob.A)
            Pm = Qr + prob.A.T @ Pm @ (prob.A - prob.B @ K)
            if it > 1 and np.linalg.norm(K - Kp) < 1e-10:
                break
        return K, Pm
    (K0, P0), (K1, P1) = lqr(prob.rho), lqr(prob.rho + h)
    return (K1 - K0) / h, (P1 - P0) / h


def case_adaptive(name, prob, x0s, settings, adaptive, sens=None, solves=2, note=""):
    """Adaptive rho (admm.cpp:147-174): per instance, `solves` consecutive solves of one solver (the adapted
    cache persists, the workspace warm-starts).  Runs on the zero-initialised build of the snapshot ("refa",
    oracle/Makefile).  sens=None: the tables tiny_setup hard-codes (12x4 only), stored as inputs."""
    inst = []
    for b in range(x0s.shape[1]):
        s = CpuSolver("refa", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        s.update_settings(**settings)
        if prob.has_bounds():
            s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if sens is None:
            sens = s.get_builtin_sensitivity()
        s.set_sensitivity(*sens)
        s.set_adaptive_rho(1, adaptive["rho_min"], adaptive["rho_max"], adaptive["clip"])
        s.set_x0(x0s[:, b])
        seq = []
        for _ in range(solves):
            st = s.solve()
            o = _sol(s, st)
            a = s.get_adapted()
            o.update(rho=a["rho"], Kinf=_l(a["Kinf"]), Pinf=_l(a["Pinf"]))
            seq.append(o)
        inst.append(seq)
    return dict(case=name, note=note, problem=_prob_dict(prob), settings=settings, adaptive=adaptive,
                dKinf_drho=_l(sens[0]), dPinf_drho=_l(sens[1]), x0=_l(x0s), batch=int(x0s.shape[1]), expect=inst)


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    build(port=False, ref=True)
    os.makedirs(OUT, exist_ok=True)
    fixed100 = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
    tol = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    cases = []

    # G1: config 1, examples/cartpole_example_one_solve.jl (N=20, rho=1, max_iter=10, no bounds)
    cases.append(case_single("G1_cartpole_one_solve", P.cartpole(20), [0.5, 0, 0, 0],
                             dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=10, check_termination=1),
                             note="BASELINE config 1; SURVEY 8(c) known-answer: 7 iterations"))
    # G2: cartpole u in [-0.5,0.5], 100 fixed iterations, cold; 8 seeded instances of config 2 + the survey x0
    x0s = np.concatenate([np.array([[0.5, 0, 0, 0.0]]).T, P.cartpole_x0(8, seed=0)], axis=1)
    cases.append(case_batch("G2_cartpole_box_fixed100", P.cartpole(20, u_bound=0.5), x0s, fixed100,
                            note="BASELINE config 2 at batch 9 (instance 0 = survey known-answer x0)"))
    # G3: tests/test_basic.jl cases (N=10)
    cases.append(case_single("G3a_test_basic_unconstrained", P.cartpole(10), [0.5, 0, 0, 0], tol,
                             xref=np.zeros((4, 10)), uref=np.zeros((1, 9)),
                             note="tests/test_basic.jl:27-44 expects status 0"))
    pb = P.cartpole(10)
    pb.x_min, pb.x_max = np.full((4, 10), -np.inf), np.full((4, 10), np.inf)
    pb.u_min, pb.u_max = np.full((1, 9), -1.0), np.full((1, 9), 1.0)
    cases.append(case_single("G3b_test_basic_bounds", pb, [1.0, 0, 0, 0], tol,
                             xref=np.zeros((4, 10)), uref=np.zeros((1, 9)),
                             note="tests/test_basic.jl:47-69 expects status 0 and |u| <= 1"))
    # G3c/d: tests/test_settings.jl (N=2; abs_pri_tol=5.0; max_iter=1)
    cases.append(case_single("G3c_test_settings_pritol5", P.cartpole(2), [0.1, 0, 0, 0],
                             dict(abs_pri_tol=5.0, abs_dua_tol=1e-3, max_iter=100, check_termination=1),
                             xref=np.zeros((4, 2)), uref=np.zeros((1, 1)),
                             note="tests/test_settings.jl:18-32"))
    cases.append(case_single("G3d_test_settings_maxiter1", P.cartpole(2), [0.1, 0, 0, 0],
                             dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=1, check_termination=1),
                             xref=np.zeros((4, 2)), uref=np.zeros((1, 1)),
                             note="tests/test_settings.jl:78-92"))
    # G4: state bound x1 in [-2,2] (others +-Inf), u +-5, to convergence
    pc = P.cartpole(20)
    pc.x_min, pc.x_max = np.full((4, 20), -np.inf), np.full((4, 20), np.inf)
    pc.x_min[0, :], pc.x_max[0, :] = -2.0, 2.0
    pc.u_min, pc.u_max = np.full((1, 19), -5.0), np.full((1, 19), 5.0)
    cases.append(case_single("G4_cartpole_state_bound", pc, [0.0, 0, 0.1, 0], tol,
                             xref=np.zeros((4, 20)), uref=np.zeros((1, 19)),
                             note="examples/cartpole_example_reference_constrained.jl:16-27"))
    pc2 = P.cartpole(20)
    pc2.x_min, pc2.x_max = np.full((4, 20), -1e17), np.full((4, 20), 1e17)
    pc2.x_min[0, :], pc2.x_max[0, :] = -0.4, 0.4
    pc2.u_min, pc2.u_max = np.full((1, 19), -5.0), np.full((1, 19), 5.0)
    cases.append(case_single("G4b_cartpole_state_bound_active", pc2, [0.39, 0.9, 0.0, 0.0], fixed100,
                             note="state bound that is active (cart pushed into the |x1| <= 0.4 wall)"))
    # G5: warm-start MPC sequence, examples/cartpole_example_mpc.jl:35-51 (max_iter=10)
    cases.append(case_mpc("G5_cartpole_mpc_warm", P.cartpole(20), [0.5, 0, 0, 0],
                          dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=10, check_termination=1),
                          steps=8, note="workspace persists between solves (SURVEY 3.5)"))
    cases.append(case_mpc("G5b_cartpole_mpc_warm_bounded", P.cartpole(20, u_bound=0.5), [0.5, 0, 0, 0],
                          dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=10, check_termination=1),
                          steps=8, note="same loop with u in [-0.5, 0.5]"))
    # G6: quadrotor N=30 rho=5 u+-0.5, seeded x0, 100 fixed its
    cases.append(case_batch("G6_quadrotor_box_fixed100", P.quadrotor(30), P.quadrotor_x0(6, seed=1),
                            fixed100, note="BASELINE config 3 at batch 6"))
    cases.append(case_batch("G6b_quadrotor_tol", P.quadrotor(30), P.quadrotor_x0(6, seed=3),
                            dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1),
                            note="config 5's tolerance-terminated variant at batch 6 (per-instance iter)"))
    # G7: rocket A,B,Q,R,box N=50 WITHOUT fdyn/SOC (box-only sub-problem), tracking refs
    xr, ur = P.rocket_refs(50)
    cases.append(case_batch("G7_rocket_box_fixed100", P.rocket(50), P.rocket_x0(4, seed=2), fixed100,
                            xref=xr, uref=ur, note="box-only sub-problem of config 4 (no fdyn, no SOC)"))
    # G8: per-iteration traces
    cases.append(case_trace("G8a_cartpole_trace", P.cartpole(20, u_bound=0.5), [0.5, 0, 0, 0], 100))
    cases.append(case_trace("G8b_quadrotor_trace", P.quadrotor(30), P.quadrotor_x0(1, seed=1)[:, 0], 60))

    # G9: adaptive rho
    ad = dict(rho_min=0.1, rho_max=10.0, clip=True)
    cases.append(case_adaptive("G9a_quadrotor_adaptive_fixed100", P.quadrotor(30), P.quadrotor_x0(4, seed=1), fixed100,
                               ad, note="config 3 with adaptive_rho, the reference's built-in 12x4 tables, 2 solves"))
    cases.append(case_adaptive("G9b_quadrotor_adaptive_tol", P.quadrotor(30), P.quadrotor_x0(4, seed=3), tol, ad,
                               note="tolerance-terminated, per-instance iteration counts"))
    pc9 = P.cartpole(20, u_bound=0.5)
    cases.append(case_adaptive("G9c_cartpole_adaptive", pc9, P.cartpole_x0(4, seed=0), fixed100,
                               dict(rho_min=0.5, rho_max=4.0, clip=True), sens=_fd_sensitivity(pc9),
                               note="config 2 with adaptive_rho, finite-difference sensitivities as input"))
    cases.append(case_adaptive("G9d_cartpole_adaptive_noclip", pc9, P.cartpole_x0(2, seed=5), fixed100,
                               dict(rho_min=0.5, rho_max=4.0, clip=False), sens=_fd_sensitivity(pc9), solves=1,
                               note="clipping off"))

    for c in cases:
        if only and not c["case"].startswith(only):
            continue
        path = os.path.join(OUT, c["case"] + ".json")
        with open(path, "w") as f:
            json.dump(c, f, allow_nan=True)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
