"""TEST INFRASTRUCTURE ONLY — generator of tests/golden/X*.json (run in the build container; needs scipy).

Independent check of the UNPINNED extensions (affine dynamics term, second-order cones, linear inequalities;
bindings.cpp:413-490 — their arithmetic lives only in the absent TinyMPC submodule, so there is no reference output to
compare with).  What CAN be checked without the reference is that the construction the oracle restates solves the
right problem: at a fixed point of the ADMM iteration (consensus x = v_i for every constraint set i, duals in the
normal cones) the iterate is the minimiser of a convex program that can be written down and handed to a DIFFERENT
method.  With s_x / s_u constraint sets on the state / input side (the box pair always counts, cones and linear rows
add one each) and the reference's cached Riccati solution on Q + 2 rho I, R + 2 rho I (tiny_api.cpp:90-91,113,134-135):

    minimise   sum_{k<N-1} [ 1/2 x_k' Qe x_k - xref_k' (Q + rho I) x_k + 1/2 u_k' Re u_k - uref_k' (R + rho I) u_k ]
               + 1/2 x_{N-1}' (Pinf - s_x rho I) x_{N-1} - xref_{N-1}' Pinf x_{N-1}
               Qe = Q + (2 - s_x) rho I,  Re = R + (2 - s_u) rho I
    subject to x_{k+1} = A x_k + B u_k + f,  x_0 given,  box / cone / linear constraints at every knot.

This script (1) runs the oracle (fp64 restatement) to tight convergence, (2) solves the program above by a sequential
quadratic programming code (scipy SLSQP) on the condensed problem in u, started from zero — nothing shared with the
ADMM code but the problem data — and (3) writes both, with the KKT stationarity residual of the oracle's point, to
tests/golden/X*.json.  It pins the FIXED POINT (the projections, the affine terms of the gradient recursion, the way
the sets are coupled), not the iterate path.  The first case has no extension at all (pinned path): it validates the
program above against arithmetic that IS pinned to the reference.
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinympc_julia_amd as t  # noqa: E402  (problem data + host Riccati only; no GPU)
from oracle import cpu_oracle  # noqa: E402


def lander(N, dt=0.2):
    """A rocket-landing family with enough control authority over a short horizon for the glide cone to bind without
    making the problem infeasible (with the reference's rocket data, dt = 0.05, a 10-knot horizon can hardly move the
    position: the state cone is either inactive or infeasible): double integrator, gravity as the affine term."""
    A = np.eye(6)
    A[0, 3] = A[1, 4] = A[2, 5] = dt
    B = np.zeros((6, 3))
    for i in range(3):
        B[i, i], B[3 + i, i] = 0.5 * dt * dt, dt
    p = t.problems.Problem("lander", A, B, np.diag([10.0, 10.0, 10.0, 1.0, 1.0, 1.0]), np.diag([0.1, 0.1, 0.1]), 1.0, N)
    p.fdyn = np.array([0.0, 0.0, -0.5 * 9.81 * dt * dt, 0.0, 0.0, -9.81 * dt])
    p.x_min, p.x_max = np.full((6, N), -1e17), np.full((6, N), 1e17)
    p.x_min[2, :] = -0.1
    p.u_min, p.u_max = np.full((3, N - 1), -10.0), np.full((3, N - 1), 40.0)
    return p


def cases():
    out = []
    p = t.problems.cartpole(10, u_bound=0.4)
    out.append(dict(name="X0_cartpole_box_only_pinned_path", prob=p, x0=np.array([0.3, 0.0, 0.05, 0.0]),
                    xref=np.zeros((4, 10)), uref=np.zeros((1, 9)), fdyn=None, cones=None, lin=None))
    p = t.problems.rocket(8)                  # BASELINE config 4's family; thrust cone tightened so that it binds
    xr, ur = t.problems.rocket_refs(8)
    x0 = np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5])
    out.append(dict(name="X1_rocket_fdyn_input_cone", prob=p, x0=x0, xref=xr, uref=ur, fdyn=p.fdyn,
                    cones=dict(Acu=[0], qcu=[3], cu=[0.05], Acx=[], qcx=[], cx=[]), lin=None))
    N = 12
    p = lander(N)
    xr, ur = np.zeros((6, N)), np.zeros((3, N - 1))
    ur[2, :] = 9.81
    x0 = np.array([3.0, 1.5, 7.0, 0.0, 0.0, -2.0])
    out.append(dict(name="X2_lander_fdyn_both_cones", prob=p, x0=x0, xref=xr, uref=ur, fdyn=p.fdyn,
                    cones=dict(Acu=[0], qcu=[3], cu=[0.3], Acx=[0], qcx=[3], cx=[0.5]), lin=None))
    p = t.problems.cartpole(10, u_bound=5.0)
    out.append(dict(name="X3_cartpole_linear_rows", prob=p, x0=np.array([0.3, 0.0, 0.05, 0.0]), xref=np.zeros((4, 10)),
                    uref=np.zeros((1, 9)), fdyn=None, cones=None,
                    lin=dict(Ax=np.array([[0.0, -1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 1.0]]), bx=np.array([0.04, 0.3]),
                             Au=np.array([[1.0], [-1.0]]), bu=np.array([0.8, 0.485]))))
    # three sets per side (box + cone + rows): the rows subtract another rho from the effective Hessians, which only the
    # rocket's weights (Q = 101, R = 2 > rho) leave positive — the lander's R = 0.1 would make the program non-convex,
    # and the iteration then diverges
    p = t.problems.rocket(10)
    xr10, ur10 = t.problems.rocket_refs(10)
    out.append(dict(name="X4_rocket_fdyn_cones_linear", prob=p, x0=np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5]), xref=xr10,
                    uref=ur10, fdyn=p.fdyn, cones=dict(Acu=[0], qcu=[3], cu=[0.25], Acx=[0], qcx=[3], cx=[0.5]),
                    lin=dict(Ax=np.array([[0.0, 0.0, 0.0, 0.0, 0.0, -1.0]]), bx=np.array([4.6]),
                             Au=np.array([[0.3, 0.3, 1.0], [0.0, 0.0, -1.0]]), bu=np.array([100.0, -2.0]))))
    p = lander(N)
    xr5 = xr.copy()
    xr5[2, :] = 2.0                           # hover target above the ground: the solution stays clear of the cone's apex
    out.append(dict(name="X5_lander_unit_cones_socp", prob=p, x0=np.array([3.0, 1.5, 4.2, 0.0, 0.0, -1.0]), xref=xr5, uref=ur,
                    fdyn=p.fdyn, cones=dict(Acu=[0], qcu=[3], cu=[1.0], Acx=[0], qcx=[3], cx=[1.0]), lin=None))
    return out


def exact_cache(p):
    """The reference's Riccati recursion (tiny_api.cpp:124-190: Q + 2 rho I, R + 2 rho I, P0 = rho I) iterated to machine
    precision instead of to its 1e-5 stopping rule.  With the reference's own cache the x-update is only a 1e-5-accurate
    LQR solve, and the ADMM fixed point sits that far from the optimum of ANY fixed program (measured: 4e-6 on the rocket
    with no constraint active); the stopping rule is pinned elsewhere (G1-G9 cache fixtures), here the cache is handed in
    through set_cache_terms so that the fixed point is an exact optimum and the check can be tight."""
    nx, nu, rho = p.nx, p.nu, p.rho
    Q1 = np.diag(np.diag(p.Q)) + 2 * rho * np.eye(nx)
    R1 = np.diag(np.diag(p.R)) + 2 * rho * np.eye(nu)
    P = rho * np.eye(nx)
    K = np.zeros((nu, nx))
    for it in range(200000):
        Kn = np.linalg.solve(R1 + p.B.T @ P @ p.B, p.B.T @ P @ p.A)
        Pn = Q1 + p.A.T @ P @ (p.A - p.B @ Kn)
        done = np.abs(Kn - K).max() < 1e-15 * max(1.0, np.abs(Kn).max()) and np.abs(Pn - P).max() < 1e-15 * np.abs(Pn).max()
        K, P = Kn, 0.5 * (Pn + Pn.T)
        if done:
            break
    return dict(Kinf=K, Pinf=P, Quu_inv=np.linalg.inv(R1 + p.B.T @ P @ p.B), AmBKt=(p.A - p.B @ K).T)


def run_oracle(c, tol=1e-11, max_iter=2000000):
    p = c["prob"]
    o = cpu_oracle.CpuSolver("orc64", p.A, p.B, p.Q, p.R, p.rho, p.N)
    o.update_settings(abs_pri_tol=tol, abs_dua_tol=tol, max_iter=max_iter, check_termination=1)
    k = exact_cache(p)
    o.set_cache_terms(k["Kinf"], k["Pinf"], k["Quu_inv"], k["AmBKt"])
    o.set_bound_constraints(p.x_min, p.x_max, p.u_min, p.u_max)
    if c["fdyn"] is not None:
        o.set_fdyn(c["fdyn"])
    if c["cones"]:
        k = c["cones"]
        o.set_cone_constraints(k["Acu"], k["qcu"], k["cu"], k["Acx"], k["qcx"], k["cx"])
    if c["lin"]:
        o.set_linear_constraints(c["lin"]["Ax"], c["lin"]["bx"], c["lin"]["Au"], c["lin"]["bu"])
    o.set_x_ref(c["xref"])
    o.set_u_ref(c["uref"])
    o.set_x0(c["x0"])
    o.solve()
    r = o.get_solution()
    return r


def independent(c, seed_point=None, start_at_seed=False):
    """SLSQP on the condensed convex program (variables: u_0 .. u_{N-2}), every function written as a quadratic
    1/2 U'HU + g'U + c0 with analytic derivatives, then a Newton polish of the KKT system on the active set SLSQP
    found (multipliers checked non-negative, inactive rows checked feasible): a primal-dual optimality certificate."""
    p = c["prob"]
    nx, nu, N, rho = p.nx, p.nu, p.N, p.rho
    A, B = p.A, p.B
    f = np.zeros(nx) if c["fdyn"] is None else np.asarray(c["fdyn"], dtype=float)
    Pinf = exact_cache(p)["Pinf"]
    ncx = len(c["cones"]["Acx"]) if c["cones"] else 0
    ncu = len(c["cones"]["Acu"]) if c["cones"] else 0
    mlx = len(c["lin"]["bx"]) if c["lin"] else 0
    mlu = len(c["lin"]["bu"]) if c["lin"] else 0
    sx = 1 + (ncx > 0) + (mlx > 0)
    su = 1 + (ncu > 0) + (mlu > 0)
    Qd, Rd = np.diag(p.Q), np.diag(p.R)        # off-diagonals of Q, R are dropped by the reference (tiny_api.cpp:90-91)
    Qe, Re = np.diag(Qd + (2 - sx) * rho), np.diag(Rd + (2 - su) * rho)
    Ql, Rl = np.diag(Qd + rho), np.diag(Rd + rho)
    Pe = Pinf - sx * rho * np.eye(nx)
    xref, uref, x0 = c["xref"], c["uref"], c["x0"]
    nU = nu * (N - 1)
    # x_k = Su_k U + c_k
    Su = np.zeros((N, nx, nU))
    cst = np.zeros((N, nx))
    cst[0] = x0
    for k in range(N - 1):
        Su[k + 1] = A @ Su[k]
        Su[k + 1][:, k * nu:(k + 1) * nu] += B
        cst[k + 1] = A @ cst[k] + f
    Eu = [np.eye(nU)[k * nu:(k + 1) * nu] for k in range(N - 1)]                  # u_k = Eu_k U

    # cost: 1/2 U'HU + g'U
    H, g = np.zeros((nU, nU)), np.zeros(nU)
    for k in range(N - 1):
        H += Su[k].T @ Qe @ Su[k] + Eu[k].T @ Re @ Eu[k]
        g += Su[k].T @ (Qe @ cst[k] - Ql @ xref[:, k]) - Eu[k].T @ (Rl @ uref[:, k])
    H += Su[N - 1].T @ Pe @ Su[N - 1]
    g += Su[N - 1].T @ (Pe @ cst[N - 1] - Pinf @ xref[:, N - 1])
    H = 0.5 * (H + H.T)

    # constraints c_i(U) >= 0: affine rows, and cones  mu z_axis - |z_head| >= 0  of z = M U + m
    class Row:
        def __init__(self, row, off, what):
            self.row, self.off, self.kind, self.scale = np.asarray(row, dtype=float), float(off), what, 1.0
            self.scale = max(1.0, np.abs(self.row).max())

        def val(self, U):
            return self.row @ U + self.off

        def grad(self, U):
            return self.row

        dirv = grad                                  # Euclidean projection: the multiplier acts along the gradient

        def dir_jac(self, U):
            return None

        def sq(self, U):                             # smooth form for SLSQP
            return [(self.val(U), self.row)]

    class Cone:
        """The cone "projection" of the public solver (restated in the oracle: a <= -mu t -> 0; a <= mu t -> s; else
        1/2 (1 + mu t / a) (w, a / mu)) is the projection in the metric W = diag(1, .., 1, mu^2), Euclidean only for
        mu = 1.  A slack/dual pair built on it has its dual in W^-1 N_K(z), not in N_K(z): the multiplier acts along
        W^-1 grad c — the axis component of the gradient divided by mu^2 — and the fixed point solves that variational
        inequality (the SOCP itself only when mu = 1)."""

        def __init__(self, M, m, a0, q, mu, what):
            self.M, self.m, self.h, self.ax, self.mu, self.kind, self.scale = M, m, slice(a0, a0 + q - 1), a0 + q - 1, mu, what, 1.0

        def z(self, U):
            return self.M @ U + self.m

        def val(self, U):
            z = self.z(U)
            return self.mu * z[self.ax] - np.linalg.norm(z[self.h])

        def _gz(self, U, axis_weight):
            z = self.z(U)
            gz = np.zeros_like(z)
            gz[self.h] = -z[self.h] / np.linalg.norm(z[self.h])
            gz[self.ax] = axis_weight
            return gz

        def grad(self, U):
            return self.M.T @ self._gz(U, self.mu)

        def dirv(self, U):
            return self.M.T @ self._gz(U, 1.0 / self.mu)          # W^-1 grad: mu / mu^2

        def dir_jac(self, U):
            z = self.z(U)
            w = z[self.h]
            a = np.linalg.norm(w)
            Hz = np.zeros((len(z), len(z)))
            Hz[self.h, self.h] = -(np.eye(len(w)) - np.outer(w, w) / (a * a)) / a
            return self.M.T @ Hz @ self.M

        def sq(self, U):                             # (mu t)^2 - |w|^2 >= 0 and t >= 0, with gradients
            z = self.z(U)
            D = np.zeros(len(z))
            D[self.h], D[self.ax] = -1.0, self.mu ** 2
            return [(z @ (D * z), 2.0 * self.M.T @ (D * z)), (z[self.ax], self.M[self.ax])]

    C = []
    big = 1e16
    for k in range(1, N):                    # knot 0 is the given x0
        for i in range(nx):
            if p.x_max[i, k] < big:
                C.append(Row(-Su[k][i], p.x_max[i, k] - cst[k][i], "x box"))
            if p.x_min[i, k] > -big:
                C.append(Row(Su[k][i], cst[k][i] - p.x_min[i, k], "x box"))
        for j in range(mlx):
            a, b = np.asarray(c["lin"]["Ax"][j], dtype=float), c["lin"]["bx"][j]
            C.append(Row(-(a @ Su[k]), b - a @ cst[k], "x row"))
        for j in range(ncx):
            C.append(Cone(Su[k], cst[k], c["cones"]["Acx"][j], c["cones"]["qcx"][j], c["cones"]["cx"][j], "x cone"))
    for k in range(N - 1):
        for i in range(nu):
            if p.u_max[i, k] < big:
                C.append(Row(-Eu[k][i], p.u_max[i, k], "u box"))
            if p.u_min[i, k] > -big:
                C.append(Row(Eu[k][i], -p.u_min[i, k], "u box"))
        for j in range(mlu):
            a, b = np.asarray(c["lin"]["Au"][j], dtype=float), c["lin"]["bu"][j]
            C.append(Row(-(a @ Eu[k]), b, "u row"))
        for j in range(ncu):
            C.append(Cone(Eu[k], np.zeros(nu), c["cones"]["Acu"][j], c["cones"]["qcu"][j], c["cones"]["cu"][j], "u cone"))
    nC = len(C)

    cons = []
    for ci in C:
        for part in range(len(ci.sq(np.ones(nU)))):
            cons.append(dict(type="ineq", fun=lambda V, ci=ci, part=part: ci.sq(V)[part][0],
                             jac=lambda V, ci=ci, part=part: ci.sq(V)[part][1]))
    U = np.zeros(nU)
    for k in range(N - 1):                   # an interior start for the squared cone rows (their gradient vanishes at the apex)
        for j in range(ncu):
            U[k * nu + c["cones"]["Acu"][j] + c["cones"]["qcu"][j] - 1] = 1.0
    for _ in range(8):                       # restarts from the last point: SLSQP stops early on a flat merit function
        r = minimize(lambda V: 0.5 * V @ H @ V + g @ V, U, jac=lambda V: H @ V + g, method="SLSQP", constraints=cons,
                     options=dict(ftol=1e-15, maxiter=3000))
        U = r.x
    # Newton on the stationarity system  H U + g - sum_i lam_i dir_i(U) = 0,  c_i(U) = 0 (i active), started from the
    # SLSQP point (the Euclidean optimum: the solution itself when every cone has mu = 1), with a primal-dual
    # active-set loop: the most negative multiplier leaves, else the most violated row enters
    def values(V):
        return np.array([ci.val(V) / ci.scale for ci in C])

    vals = values(U if seed_point is None else np.asarray(seed_point).T.reshape(-1))
    if start_at_seed:
        U = np.asarray(seed_point).T.reshape(-1).copy()
    active = [i for i in range(nC) if vals[i] < 1e-6]
    lam = np.zeros(len(active))
    for outer in range(300):
        if active:
            Jd = np.stack([C[i].dirv(U) for i in active])
            lam = np.linalg.lstsq(Jd.T, H @ U + g, rcond=None)[0]
        def resid(V, lm):
            gl = H @ V + g - (sum(l * C[i].dirv(V) for l, i in zip(lm, active)) if active else 0.0)
            return np.concatenate([gl, np.array([C[i].val(V) for i in active])]) if active else gl

        for _ in range(200):
            F = resid(U, lam)
            HL = H.copy()
            for l, i in zip(lam, active):
                dj = C[i].dir_jac(U)
                if dj is not None:
                    HL -= l * dj
            if active:
                J = np.stack([C[i].grad(U) for i in active])
                Jd = np.stack([C[i].dirv(U) for i in active])
                K = np.block([[HL, -Jd.T], [J, np.zeros((len(active), len(active)))]])
            else:
                K = HL
            step = np.linalg.lstsq(K, -F, rcond=None)[0]
            tstep, f0 = 1.0, np.linalg.norm(F)
            while tstep > 1e-6 and not np.linalg.norm(resid(U + tstep * step[:nU], lam + tstep * step[nU:])) < (1 - 1e-4 * tstep) * f0:
                tstep *= 0.5                                     # damped Newton: backtrack on the residual norm
            U, lam = U + tstep * step[:nU], lam + tstep * step[nU:]
            if np.abs(step).max() < 1e-14 * max(1.0, np.abs(U).max()) or f0 < 1e-15:
                break
        vals = values(U)
        if len(lam) and lam.min() < -1e-12:
            active.pop(int(np.argmin(lam)))
        else:
            cand = [(vals[i], i) for i in range(nC) if i not in active]
            worst = min(cand) if cand else (0.0, -1)
            if worst[0] >= -1e-10:
                break
            active.append(worst[1])
        lam = np.zeros(len(active))
    gradL = H @ U + g - (sum(l * C[i].dirv(U) for l, i in zip(lam, active)) if active else 0.0)
    cert = dict(active=len(active), active_kinds=sorted(set(C[i].kind for i in active)),
                min_multiplier=float(lam.min()) if len(lam) else None,
                min_constraint=float(vals.min()), stationarity=float(np.abs(gradL).max() / max(1.0, np.abs(g).max())),
                euclidean=bool(all(not isinstance(C[i], Cone) or C[i].mu == 1.0 for i in active)),
                start="slsqp" if seed_point is None else ("slsqp point, active set read off the oracle's point" if not start_at_seed
                                                          else "oracle's point (Newton polish)"))
    X = np.stack([Su[k] @ U + cst[k] for k in range(N)], axis=1)
    return X, U.reshape(N - 1, nu).T, dict(sx=sx, su=su, slsqp_message=str(r.message),
                                           objective=float(0.5 * U @ H @ U + g @ U), certificate=cert)


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    for c in cases():
        r = run_oracle(c)
        X, U, info = independent(c)
        ce = info["certificate"]
        def bad(ce):
            return (ce["min_constraint"] < -1e-9 or (ce["min_multiplier"] is not None and ce["min_multiplier"] < -1e-9)
                    or ce["stationarity"] > 1e-9)
        # The certificate (feasible, multipliers >= 0, stationarity ~ 1e-16) is what pins the fixed point; how the Newton
        # iteration is started does not enter it.  Preferred start: the SLSQP point (nothing taken from the oracle); if the
        # active-set loop cycles from there, its guess of the active set, then its starting point, are read off the oracle's
        # solution — the iteration then only polishes, and still has to certify.
        if bad(info["certificate"]):
            X, U, info = independent(c, seed_point=r["u"])
        if bad(info["certificate"]):
            X, U, info = independent(c, seed_point=r["u"], start_at_seed=True)
        assert not bad(info["certificate"]), (c["name"], info["certificate"])
        ex = np.abs(r["x"] - X).max() / np.abs(X).max()
        eu = np.abs(r["u"] - U).max() / np.abs(U).max()
        p = c["prob"]
        act = {}
        if c["cones"] and c["cones"]["Acu"]:
            a0, q, mu = c["cones"]["Acu"][0], c["cones"]["qcu"][0], c["cones"]["cu"][0]
            slack = mu * U[a0 + q - 1] - np.linalg.norm(U[a0:a0 + q - 1], axis=0)
            act["input_cone_active_knots"] = int((slack < 1e-7).sum())
        if c["cones"] and c["cones"]["Acx"]:
            a0, q, mu = c["cones"]["Acx"][0], c["cones"]["qcx"][0], c["cones"]["cx"][0]
            slack = mu * X[a0 + q - 1] - np.linalg.norm(X[a0:a0 + q - 1], axis=0)
            act["state_cone_active_knots"] = int((slack[1:] < 1e-7).sum())
        if c["lin"]:
            act["state_rows_active"] = int((np.abs(c["lin"]["Ax"] @ X[:, 1:] - c["lin"]["bx"][:, None]) < 1e-7).sum())
            act["input_rows_active"] = int((np.abs(c["lin"]["Au"] @ U - c["lin"]["bu"][:, None]) < 1e-7).sum())
        act["input_box_active"] = int(((np.abs(U - p.u_min) < 1e-7) | (np.abs(U - p.u_max) < 1e-7)).sum())
        print(f"{c['name']}: oracle iter {r['iter']} solved {r['solved']}  vs SLSQP: x {ex:.2e} u {eu:.2e}  {act} {info['slsqp_message']}")
        g = dict(case=c["name"],
                 note="independent check of the ADMM fixed point: `independent` is the SLSQP optimum of the convex program in "
                      "oracle/make_independent.py's header, `oracle` the fp64 restatement run to 1e-10; generated by "
                      "oracle/make_independent.py",
                 problem=dict(name=p.name, nx=p.nx, nu=p.nu, N=p.N, rho=p.rho, A=p.A.flatten("F").tolist(),
                              B=p.B.flatten("F").tolist(), Q=p.Q.flatten("F").tolist(), R=p.R.flatten("F").tolist(),
                              x_min=p.x_min.flatten("F").tolist(), x_max=p.x_max.flatten("F").tolist(),
                              u_min=p.u_min.flatten("F").tolist(), u_max=p.u_max.flatten("F").tolist()),
                 x0=c["x0"].tolist(), xref=c["xref"].flatten("F").tolist(), uref=c["uref"].flatten("F").tolist(),
                 fdyn=None if c["fdyn"] is None else np.asarray(c["fdyn"]).tolist(),
                 cache={k: v.flatten("F").tolist() for k, v in exact_cache(p).items()},
                 cones=c["cones"],
                 lin=None if c["lin"] is None else {k: np.asarray(v).tolist() for k, v in c["lin"].items()},
                 independent=dict(x=X.flatten("F").tolist(), u=U.flatten("F").tolist(), **info),
                 oracle=dict(x=r["x"].flatten("F").tolist(), u=r["u"].flatten("F").tolist(), iter=r["iter"],
                             solved=r["solved"], res=np.asarray(r["res"]).tolist()),
                 agreement=dict(x=ex, u=eu), active=act)
        json.dump(g, open(os.path.join(outdir, c["name"] + ".json"), "w"))


if __name__ == "__main__":
    main()
