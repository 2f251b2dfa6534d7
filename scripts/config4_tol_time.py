"""config 4 (rocket N=50, cones + affine term, batch 32 768) with the termination check live every iteration (the reference
example's settings, rocket_landing_constraints.jl:61-62) against the fixed-iteration benchmark form"""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
B, N = 32768, int(os.environ.get("N", 50))
prob = t.problems.rocket(N); x0 = t.problems.rocket_x0(B, seed=2); xr, ur = t.problems.rocket_refs(N)
for label, kw in (("fixed 100", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)),
                  ("tol 2e-3/1e-3, check every iteration", dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)),
                  ("tol 2e-3/1e-3, check every 10", dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=10))):
    for env in (None, "1"):
        if env: os.environ["TINYMPC_HIP_NO_MFMAC"] = "1"
        else: os.environ.pop("TINYMPC_HIP_NO_MFMAC", None)
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(**kw)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        bs.set_fdyn(prob.fdyn); bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
        bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0); bs.set_profiling(True)
        for _ in range(5): bs.solve()
        st = bs.get_status()
        print(f"{label:40s} {bs.kernel_name:14s} {bs.kernel_elapsed_ms(3):7.3f} ms  mean iters {st['iter'].mean():.1f}")
        bs.close()
