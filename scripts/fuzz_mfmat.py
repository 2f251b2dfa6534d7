"""Randomised parity sweep of the transposed-sets kernel (mfmat) against the fp64 oracle: random stable (6,3) families at the
compiled horizons, the compiled cone layout on either / both / neither side, per-knot or constant or no state bounds, zero or
shared references, with / without the affine term, fixed-iteration and tolerance-terminated settings, and — per case — one of
three calling patterns: cold one-shot; TWO consecutive solves with the workspace kept (second x0 = the plant's next state),
each compared with a persistent oracle including the workspace; a fused closed loop of a few steps against the oracle loop.
Usage: python scripts/fuzz_mfmat.py [first_seed] [n_cases]
PRECISION=2 in the environment runs every case with tinympc_set_precision(s, 2) (fp64 end to end, generic kernel) at the plain
1e-5 of the suite — the closed-loop pattern then stepped from the host (precision 2 has no fused loop)."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
from oracle import cpu_oracle
from tests.util import parity_every_instance, nrel, FP32_TOL


def one(seed, precision=0, tol=None):
    tol = tol if tol is not None else (1e-5 if precision == 2 else 2e-5)      # solution
    tol_ws, tol_step = (tol, tol) if precision == 2 else (4e-5, 3e-5)            # workspace arrays / closed-loop steps
    rng = np.random.default_rng(seed)
    nx, nu = 6, 3
    N = int(rng.choice([10, 20, 30, 50]))
    B = int(rng.integers(5, 50))
    A = np.eye(nx) + 0.2 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= rng.uniform(0.9, 0.99) / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)),
                              np.diag(rng.uniform(0.5, 3.0, nu)), float(rng.uniform(0.5, 2.0)), N)
    prob.x_min, prob.x_max = -rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N)), rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N))
    prob.u_min, prob.u_max = -rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1)), rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1))
    if rng.random() < 0.35:                                # per-knot bounds: the BV variants
        prob.x_min[:, N // 2:] -= 0.3
        prob.u_max[:, ::2] += 0.1
    if rng.random() < 0.25:
        prob.x_min[:], prob.x_max[:] = -1e17, 1e17
    fdyn = 0.02 * rng.standard_normal(nx) if rng.random() < 0.7 else None
    refs = rng.random() < 0.6
    xr = 0.2 * rng.standard_normal((nx, N)) if refs else None
    ur = 0.1 * rng.standard_normal((nu, N - 1)) if refs else None
    cu = ([0], [3], [float(rng.uniform(0.3, 1.2))]) if rng.random() < 0.6 else ([], [], [])
    cx = ([0], [3], [float(rng.uniform(0.3, 1.5))]) if rng.random() < 0.6 else ([], [], [])
    cones = (cu[0], cu[1], cu[2], cx[0], cx[1], cx[2]) if (cu[0] or cx[0]) else None
    kw = [dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=int(rng.integers(10, 60)), check_termination=int(rng.choice([1, 3, 7]))),
          dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=int(rng.integers(30, 90)), check_termination=int(rng.choice([1, 5, 10])))][int(rng.integers(0, 2))]
    pattern = ["one_shot", "workspace", "rollout"][int(rng.integers(0, 3))]
    if pattern == "rollout" and kw["abs_pri_tol"] == 0.0:
        kw["max_iter"] = min(kw["max_iter"], 25)
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    f = fdyn if fdyn is not None else np.zeros(nx)

    def mk(b=None):
        o = cpu_oracle.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if fdyn is not None: o.set_fdyn(fdyn)
        if cones is not None: o.set_cone_constraints(*cones)
        if xr is not None: o.set_x_ref(xr); o.set_u_ref(ur)
        return o
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if fdyn is not None: bs.set_fdyn(fdyn)
    if cones is not None: bs.set_cone_constraints(*cones)
    if xr is not None: bs.set_x_ref(xr); bs.set_u_ref(ur)
    bs.set_warm_start(pattern != "one_shot")
    bs.set_precision(precision)
    tag = f"seed {seed} N={N} B={B} {pattern} cones={cones is not None} fdyn={fdyn is not None} refs={refs} {kw}"
    ok, name = True, None
    try:
        if pattern == "one_shot":
            X, U = np.zeros((nx, N, B)), np.zeros((nu, N - 1, B))
            it, so, res = np.zeros(B, dtype=int), np.zeros(B, dtype=int), np.zeros((B, 4))
            for b in range(B):
                o = mk(); o.set_x0(x0[:, b]); o.solve(); r_ = o.get_solution()
                X[:, :, b], U[:, :, b], it[b], so[b], res[b] = r_["x"], r_["u"], r_["iter"], r_["solved"], r_["res"]
                o.close()
            bs.set_x0(x0); bs.solve(); name = bs.kernel_name
            parity_every_instance(bs.get_solution(), bs.get_status(), dict(x=X, u=U, iter=it, solved=so, res=res), mk, x0, kw, prob.rho,
                                  tol=tol, min_same=0.0, tag=tag)
        elif pattern == "workspace":
            orcs = [mk() for _ in range(B)]
            x = x0.copy()
            for k in range(2):
                bs.set_x0(x); bs.solve(); name = bs.kernel_name
                sol, st, ws = bs.get_solution(), bs.get_status(), bs.get_workspace()
                xn = np.zeros_like(x)
                for b in range(B):
                    o = orcs[b]
                    o.set_x0(x[:, b])
                    pre, pre_c = o.get_state(), o.get_cone_state()
                    o.solve(); r = o.get_solution()
                    if int(st["iter"][b]) != r["iter"]:
                        assert abs(int(st["iter"][b]) - r["iter"]) <= max(1, kw["check_termination"]), (tag, b, st["iter"][b], r["iter"])
                        o.set_state(*[pre[key] for key in ("d", "y", "g", "v", "z")]); o.set_cone_state(**pre_c)
                        o.set_forced_exit(int(st["iter"][b]) if st["solved"][b] else -1)
                        o.solve(); o.set_forced_exit(0); r = o.get_solution()
                    sv = o.get_state()
                    ex_, eu_ = nrel(sol["states"][:, :, b], r["x"]), nrel(sol["controls"][:, :, b], r["u"])
                    assert ex_ <= tol and eu_ <= tol, f"{tag} solve {k} instance {b}: x {ex_:.3e} u {eu_:.3e}"
                    for key in ("d", "y", "g", "v", "z"):
                        e_ = np.abs(ws[key][:, :, b] - sv[key]).max() / max(np.abs(sv[key]).max(), 1e-2)
                        assert e_ <= tol_ws, f"{tag} solve {k} instance {b} workspace {key}: {e_:.3e}"
                    xn[:, b] = prob.A @ x[:, b] + prob.B @ r["u"][:, 0] + f
                x = xn
            for o in orcs: o.close()
        else:
            steps = 4
            log = None
            if precision == 2:      # no fused loop: the same loop stepped from the host (x+ = A x + B u0 + f in fp64 here)
                log = dict(u=np.zeros((nu, steps, B)), x=np.zeros((nx, steps, B)), iter=np.zeros((steps, B), dtype=int), solved=np.zeros((steps, B), dtype=int))
                xh = x0.copy()
                for k in range(steps):
                    bs.set_x0(xh); bs.solve(); name = bs.kernel_name
                    u0 = bs.get_solution()["controls"][:, 0, :]
                    sth = bs.get_status()
                    xh = prob.A @ xh + prob.B @ u0 + f[:, None]
                    log["u"][:, k, :], log["x"][:, k, :], log["iter"][k], log["solved"][k] = u0, xh, sth["iter"], sth["solved"]
            else:
                bs.set_x0(x0); log = bs.mpc_rollout(steps); name = bs.kernel_name
            for b in range(B):
                o = mk(); x = x0[:, b].copy()
                for k in range(steps):
                    o.set_forced_exit(int(log["iter"][k, b]) if log["solved"][k, b] else -1)
                    o.set_x0(x); o.solve(); r = o.get_solution()
                    x = prob.A @ x + prob.B @ r["u"][:, 0] + f
                    eu_ = np.abs(log["u"][:, k, b] - r["u"][:, 0]).max() / max(np.abs(r["u"]).max(), 1e-3)
                    ex_ = np.abs(log["x"][:, k, b] - x).max() / max(np.abs(x).max(), 1e-3)
                    assert eu_ <= tol_step and ex_ <= tol_step, f"{tag} step {k} instance {b}: u0 {eu_:.3e} x+ {ex_:.3e}"
                o.close()
    except AssertionError as e:
        ok = False
        print("FAIL", tag, name, str(e)[:300], flush=True)
    bs.close()
    return name, pattern, ok


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    cpu_oracle.build(port=True, ref=False) if not os.path.isfile(cpu_oracle.PORT_LIB) else None
    tally = {}
    for seed in range(first, first + n):
        name, pattern, ok = one(seed, precision=int(os.environ.get("PRECISION", "0")))
        key = (name, pattern)
        tally.setdefault(key, [0, 0])
        tally[key][0 if ok else 1] += 1
        if (seed - first) % 10 == 9:
            print("...", seed - first + 1, "cases", flush=True)
    for k, v in sorted(tally.items(), key=lambda kv: str(kv[0])):
        print(k, "ok", v[0], "FAIL", v[1])
