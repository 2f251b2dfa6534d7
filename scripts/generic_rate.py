"""Throughput of the generic (any-shape) kernel next to the specialised one, same family, batch 65 536."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tinympc_julia_amd as t
B = 65536
for N in (19, 20):
    prob, x0 = t.problems.cartpole(N, u_bound=0.5), t.problems.cartpole_x0(B, seed=0)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_warm_start(False); bs.set_x0(x0); bs.set_profiling(True)
    bs.solve(); bs.solve()
    print(f"N={N} kernel={bs.kernel_name} kernel_ms={bs.kernel_elapsed_ms():.3f} solves/s={B/(bs.kernel_elapsed_ms()*1e-3):.3e}")
