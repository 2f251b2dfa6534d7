"""Fixed cost of a launch (prologue + epilogue: everything that does not scale with the iteration count) of the BASELINE configs:
kernel time at max_iter = 25 / 50 / 100 / 200, straight-line fit."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
P = t.problems
for name, prob, x0, extra in (("cartpole 65536", P.cartpole(20, u_bound=0.5), P.cartpole_x0(65536, 0), None),
                              ("quadrotor 65536", P.quadrotor(30, u_bound=0.5), P.quadrotor_x0(65536, 1), None),
                              ("rocket_soc 32768", P.rocket(50), P.rocket_x0(32768, 2), "soc")):
    ts = []
    for iters in (25, 50, 100, 200):
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=x0.shape[1])
        bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters, check_termination=1)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if extra:
            bs.set_fdyn(prob.fdyn); bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
            xr, ur = P.rocket_refs(50); bs.set_x_ref(xr); bs.set_u_ref(ur)
        bs.set_warm_start(False); bs.set_x0(x0); bs.set_profiling(True)
        for _ in range(10): bs.solve()
        ts.append(bs.kernel_elapsed_ms(6)); name_k = bs.last_launch_name; bs.close()
    b, a = np.polyfit([25, 50, 100, 200], ts, 1)
    print(f"{name:18s} {name_k:16s} ms at 25/50/100/200 iterations: {[round(v, 4) for v in ts]}  per iteration {b * 1e3:.2f} us, fixed part {a * 1e3:.1f} us ({100 * a / ts[2]:.1f} % of the 100-iteration launch)", flush=True)
