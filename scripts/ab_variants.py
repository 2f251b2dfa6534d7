"""A/B timing of library variants in ONE run (same box, same clocks): config 4 fixed-iteration and check-every-iteration"""
import numpy as np, sys, os, subprocess
V = sys.argv[1:]
code = r'''
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import torch
import tinympc_julia_amd as t
from tinympc_julia_amd import tinympc as tm
tm.load_library(sys.argv[1])
B, N = 32768, 50
prob = t.problems.rocket(N); x0 = t.problems.rocket_x0(B, seed=2); xr, ur = t.problems.rocket_refs(N)
out = []
for kw in (dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1), dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)):
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_fdyn(prob.fdyn); bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
    bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0); bs.set_profiling(True)
    for _ in range(12): bs.solve()
    out.append(round(bs.kernel_elapsed_ms(8), 3)); bs.close()
print(os.path.basename(sys.argv[1]), out)
'''
for rep in range(3):
    for v in V:
        subprocess.run([sys.executable, "-c", code, v], check=False)
