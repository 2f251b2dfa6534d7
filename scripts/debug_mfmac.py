import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
from oracle import cpu_oracle
N, B = int(os.environ.get("N", 3)), 1
prob = t.problems.rocket(N)
x0 = t.problems.rocket_x0(B, seed=2)
xr, ur = t.problems.rocket_refs(N)
cones = ([0], [3], [0.25], [0], [3], [0.5])
for iters in (1, 2, 3):
    kw = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters, check_termination=1)
    o = cpu_oracle.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
    o.update_settings(**kw); o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    o.set_cone_constraints(*cones); o.set_x_ref(xr); o.set_u_ref(ur); o.set_x0(x0[:, 0]); o.solve(); r = o.get_solution()
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw); bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_cone_constraints(*cones); bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0); bs.solve()
    s = bs.get_solution()
    print(bs.kernel_name, "iters", iters)
    print(" x gpu", s["states"][:, :, 0].T.round(5).tolist()); print(" x ref", r["x"].T.round(5).tolist())
    print(" u gpu", s["controls"][:, :, 0].T.round(5).tolist()); print(" u ref", r["u"].T.round(5).tolist())
    bs.close()
