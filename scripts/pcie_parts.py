import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, tinympc_julia_amd as t
B = 65536
prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B, seed=0)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
bs.set_warm_start(False)
xs, us = np.zeros(4 * 20 * B), np.zeros(19 * B)
dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
acc = {}
def tm(name, f):
    t0 = time.perf_counter(); f(); acc[name] = acc.get(name, 0) + time.perf_counter() - t0
for it in range(13):
    if it == 3: acc.clear()
    tm("set_x0", lambda: bs.lib.tinympc_set_x0(bs.h, dp(x0), B))
    tm("solve", lambda: bs.lib.tinympc_solve(bs.h))
    tm("get_states", lambda: bs.lib.tinympc_get_states(bs.h, dp(xs)))
    tm("get_controls", lambda: bs.lib.tinympc_get_controls(bs.h, dp(us)))
print({k: round(1e3 * v / 10, 3) for k, v in acc.items()})
