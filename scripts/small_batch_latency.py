import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, tinympc_julia_amd as t
for fam in ("cartpole", "quadrotor"):
    prob = t.problems.cartpole(20, u_bound=0.5) if fam == "cartpole" else t.problems.quadrotor(30, u_bound=0.5)
    x0 = (t.problems.cartpole_x0 if fam == "cartpole" else t.problems.quadrotor_x0)(1, seed=0)
    for B in (1, 64, 1024):
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        bs.set_warm_start(False); bs.set_x0(np.repeat(x0, B, axis=1))
        for _ in range(20): bs.solve()
        t0 = time.perf_counter(); n = 200
        for _ in range(n): bs.solve()
        dt = (time.perf_counter() - t0) / n
        print(f"{fam} B={B} {bs.kernel_name}: {1e6*dt:.0f} us per solve call, iters {int(bs.get_status()['iter'][0])}")
        bs.close()
