#!/usr/bin/env python3
"""Matrix-core kernel vs quad kernel on the shapes that have both (one-shot solves, fixed 100 iterations)."""
import os, subprocess, sys
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, tinympc_julia_amd as t
fam, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
prob = t.problems.quadrotor(N, u_bound=0.5) if fam == "quadrotor" else t.problems.rocket(N)
x0 = (t.problems.quadrotor_x0 if fam == "quadrotor" else t.problems.rocket_x0)(B, seed=1)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
if fam == "rocket":
    xr, ur = t.problems.rocket_refs(N); bs.set_x_ref(xr); bs.set_u_ref(ur)
bs.set_warm_start(False); bs.set_profiling(True); bs.set_x0(x0)
for _ in range(5): bs.solve()
print(fam, N, B, bs.kernel_name, "kernel_ms=%.3f" % bs.kernel_elapsed_ms(3))
'''
for fam, N, B in (("quadrotor", 30, 65536), ("quadrotor", 20, 65536), ("quadrotor", 30, 4096), ("rocket", 10, 32768), ("rocket", 10, 262144)):
    for env in ({}, {"TINYMPC_HIP_NO_MFMA": "1"}):
        subprocess.run([sys.executable, "-c", code, fam, str(N), str(B)], env={**os.environ, **env})
