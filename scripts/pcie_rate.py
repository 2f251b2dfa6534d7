"""PCIe-inclusive rate of configs[1]: host fp64 buffers in (set_x0), solve, host fp64 buffers out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tinympc_julia_amd as t
B = 65536
prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B, seed=0)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
bs.set_warm_start(False)
for _ in range(2):
    bs.set_x0(x0); bs.solve(); bs.get_solution()
t0 = time.perf_counter(); n = 5
for _ in range(n):
    bs.set_x0(x0); bs.solve(); sol = bs.get_solution()
dt = (time.perf_counter() - t0) / n
print(f"PCIe-inclusive (fp64 host buffers in/out, ctypes): {dt*1e3:.2f} ms per 65536-solve batch = {B/dt:.3e} solves/s")
