"""Where does the digit go in the one GPU parity case that misses 1e-5 (quadrotor, synthetic tight state bounds
+-0.12, unconverged after 60 iterations; tests/test_gpu_parity.py::test_matrix_core_kernel_vs_oracle[quadrotor30_refs_bounds])?
CPU emulation of the kernel's arithmetic (fp64 recurrences, fp32 state and elementwise steps) with one array at a
time promoted to fp64, against the fp64 oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinympc_julia_amd as t
from oracle import cpu_oracle
from tests.util import nrel

N, B = 30, 171
rng = np.random.default_rng(5)
prob = t.problems.quadrotor(N)
x0all = t.problems.quadrotor_x0(B, seed=4)
prob.x_min, prob.x_max = np.full((12, N), -0.12), np.full((12, N), 0.12)
prob.x_min[:, N // 2:] = -0.2
xref, uref = 0.05 * rng.standard_normal((12, N)), 0.02 * rng.standard_normal((4, N - 1))
kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60, check_termination=1)
c = t.host_precompute(prob.A, prob.B, prob.Q, prob.R, prob.rho)
Qd, Rd = np.diag(prob.Q) + prob.rho, np.diag(prob.R) + prob.rho


def emulate(b, G=np.float32, Y=np.float32, V=np.float32, D=np.float32, EL=np.float32):
    nx, nu = 12, 4
    K, P, Qi, Am, A, Bm = c["Kinf"], c["Pinf"], c["Quu_inv"], c["AmBKt"], prob.A, prob.B
    rho = EL(prob.rho)
    g = np.zeros((nx, N), G); v = np.zeros((nx, N), V); y = np.zeros((nu, N - 1), Y); z = np.zeros((nu, N - 1), V)
    d = np.zeros((nu, N - 1), D)
    x0 = x0all[:, b].astype(np.float32).astype(np.float64)
    xmin, xmax, umin, umax = (a.astype(EL) for a in (prob.x_min, prob.x_max, prob.u_min, prob.u_max))
    for it in range(kw["max_iter"]):
        x = x0.copy(); vn = np.zeros((nx, N), V); zn = np.zeros((nu, N - 1), V)
        for j in range(N):
            xf = x.astype(EL)
            w = np.minimum(xmax[:, j], np.maximum(xmin[:, j], xf + g[:, j].astype(EL)))
            g[:, j] = ((g[:, j].astype(EL) + xf) - w).astype(G); vn[:, j] = w.astype(V)
            if j < N - 1:
                u = -(K @ x) - d[:, j].astype(np.float64)
                uf = u.astype(EL)
                zz = np.minimum(umax[:, j], np.maximum(umin[:, j], uf + y[:, j].astype(EL)))
                y[:, j] = ((y[:, j].astype(EL) + uf) - zz).astype(Y); zn[:, j] = zz.astype(V)
                x = A @ x + Bm @ u
        v[:] = vn; z[:] = zn
        p = -(P @ xref[:, N - 1]) - (rho * (vn[:, N - 1].astype(EL) - g[:, N - 1].astype(EL))).astype(np.float64)
        for j in range(N - 2, -1, -1):
            r = (-(uref[:, j].astype(EL) * Rd.astype(EL)) - rho * (zn[:, j].astype(EL) - y[:, j].astype(EL))).astype(np.float64)
            q = (-(xref[:, j].astype(EL) * Qd.astype(EL)) - rho * (vn[:, j].astype(EL) - g[:, j].astype(EL))).astype(np.float64)
            d[:, j] = (Qi @ (Bm.T @ p + r)).astype(D)
            p = q + Am @ p - K.T @ r
    return vn.astype(np.float64), zn.astype(np.float64), np.abs(g).max()


ref = cpu_oracle.solve_batch("orc64", prob, x0all, xref=xref, uref=uref, **kw)
for b in (4, 0, 17):
    print(f"instance {b}: oracle iter {ref['iter'][b]}, max|x| {np.abs(ref['x'][:, :, b]).max():.3f}")
    for label, k2 in (("kernel arithmetic (all state fp32)", {}), ("state dual g in fp64", dict(G=np.float64)),
                      ("input dual y in fp64", dict(Y=np.float64)), ("slack v, z in fp64", dict(V=np.float64)),
                      ("feed-forward d in fp64", dict(D=np.float64)),
                      ("g, y in fp64 + fp64 elementwise", dict(G=np.float64, Y=np.float64, EL=np.float64)),
                      ("everything fp64 but x0", dict(G=np.float64, Y=np.float64, V=np.float64, D=np.float64, EL=np.float64))):
        X, U, gmax = emulate(b, **k2)
        print(f"   {label:38s} x err {nrel(X, ref['x'][:, :, b]):.2e}  u err {nrel(U, ref['u'][:, :, b]):.2e}   max|g| {gmax:.2f}")
