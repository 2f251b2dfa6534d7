"""CPU emulation of the kernels' arithmetic on the warm-start fixtures (G5 / G5b): which rounding costs the digit?
state dtype (d, y, g, v, z as kept in registers / the workspace), recurrence dtype, x0 dtype, elementwise dtype."""
import json, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import cm, load_golden, problem_of, nrel
import tinympc_julia_amd as t


def run(gname, S=np.float32, RT=np.float64, X0=np.float32, EL=np.float32, DT=None, verbose=False):
    DT = DT or S
    g = load_golden(gname)
    prob = problem_of(g)
    nx, nu, N = prob.nx, prob.nu, prob.N
    c = t.host_precompute(prob.A, prob.B, prob.Q, prob.R, prob.rho)
    K, P, Qi, Am = (c[k].astype(RT) for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt"))
    A, B = prob.A.astype(RT), prob.B.astype(RT)
    rho = EL(prob.rho)
    st = g["settings"]
    bounded = prob.has_bounds()
    if bounded:
        umin, umax = prob.u_min.astype(EL), prob.u_max.astype(EL)
        xmin, xmax = prob.x_min.astype(EL), prob.x_max.astype(EL)
    d = np.zeros((nu, N - 1), DT); y = np.zeros((nu, N - 1), S); z = np.zeros((nu, N - 1), S)
    gg = np.zeros((nx, N), S); v = np.zeros((nx, N), S)
    worst = []
    for k, step in enumerate(g["steps"]):
        x0 = np.array(step["x0"]).astype(X0).astype(RT)
        it, conv = 0, False
        for i in range(st["max_iter"]):
            # forward
            x = x0.copy(); vn = np.zeros((nx, N), S); zn = np.zeros((nu, N - 1), S)
            px = dx = pu = du = EL(0)
            for j in range(N):
                xf = x.astype(EL)
                w = xf + gg[:, j].astype(EL)
                if bounded: w = np.minimum(xmax[:, j], np.maximum(xmin[:, j], w))
                gnew = (gg[:, j].astype(EL) + xf) - w
                px = max(px, np.abs(xf - w).max()); dx = max(dx, np.abs(v[:, j].astype(EL) - w).max())
                gg[:, j] = gnew.astype(S); vn[:, j] = w.astype(S)
                if j < N - 1:
                    u = -(K @ x) - d[:, j].astype(RT)
                    uf = u.astype(EL)
                    zz = uf + y[:, j].astype(EL)
                    if bounded: zz = np.minimum(umax[:, j], np.maximum(umin[:, j], zz))
                    ynew = (y[:, j].astype(EL) + uf) - zz
                    pu = max(pu, np.abs(uf - zz).max()); du = max(du, np.abs(z[:, j].astype(EL) - zz).max())
                    y[:, j] = ynew.astype(S); zn[:, j] = zz.astype(S)
                    x = A @ x + B @ u
            it += 1
            if st["check_termination"] and it % st["check_termination"] == 0:
                if px < st["abs_pri_tol"] and pu < st["abs_pri_tol"] and dx * rho < st["abs_dua_tol"] and du * rho < st["abs_dua_tol"]:
                    conv = True
            if conv:
                break
            v[:] = vn; z[:] = zn
            # backward (zero refs)
            p = (-(rho * (vn[:, N - 1].astype(EL) - gg[:, N - 1].astype(EL)))).astype(RT)
            for j in range(N - 2, -1, -1):
                r = (-(rho * (zn[:, j].astype(EL) - y[:, j].astype(EL)))).astype(RT)
                q = (-(rho * (vn[:, j].astype(EL) - gg[:, j].astype(EL)))).astype(RT)
                d[:, j] = (Qi @ (B.T @ p + r)).astype(DT)
                p = q + Am @ p - K.T @ r
        ex, eu = nrel(vn, cm(step["x"], nx, N)), nrel(zn, cm(step["u"], nu, N - 1))
        ed = nrel(d, cm(step["state_after"]["d"], nu, N - 1))
        worst.append((ex, eu, ed, it, step["iter"]))
        if verbose: print(f"  step {k}: it {it}/{step['iter']} ex {ex:.2e} eu {eu:.2e} ed {ed:.2e}")
    return worst


for gname in ("G5_cartpole_mpc_warm", "G5b_cartpole_mpc_warm_bounded"):
    print(gname)
    for label, kw in (("all fp64", dict(S=np.float64, X0=np.float64, EL=np.float64)),
                      ("kernel: fp32 state/elementwise/x0, fp64 recurrences", dict()),
                      ("  + x0 fp64", dict(X0=np.float64)),
                      ("  + d fp64", dict(DT=np.float64)),
                      ("  + all state fp64 (elementwise fp32)", dict(S=np.float64)),
                      ("  + elementwise fp64 (state fp32)", dict(EL=np.float64)),
                      ("all fp32 (precision=1)", dict(RT=np.float32))):
        w = run(gname, **kw)
        print(f"{label:55s} worst x {max(a[0] for a in w):.2e}  u {max(a[1] for a in w):.2e}  d {max(a[2] for a in w):.2e}  iters ok {all(a[3]==a[4] for a in w)}")
