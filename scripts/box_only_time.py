import numpy as np, sys, os
sys.path.insert(0, "/root/repo")
import tinympc_julia_amd as t
B, N = 32768, int(os.environ.get("N", 10))
prob = t.problems.rocket(N); x0 = t.problems.rocket_x0(B, seed=2); xr, ur = t.problems.rocket_refs(N)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0); bs.set_profiling(True)
for _ in range(6): bs.solve()
print(N, bs.kernel_name, round(bs.kernel_elapsed_ms(3), 3), "ms")
