"""kernel time of the LDS-resident matrix-core kernel over the horizon (LDS per wavefront ~ N): how many wavefronts a CU
holds, and the time per knot step"""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
B = int(os.environ.get("B", 32768))
for N in [int(a) for a in os.environ.get("NS", "50,40,30,25,20,12").split(",")]:
    for mode in ("fdyn+cones", "fdyn"):
        prob = t.problems.rocket(N)
        x0 = t.problems.rocket_x0(B, seed=2)
        xr, ur = t.problems.rocket_refs(N)
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        bs.set_fdyn(prob.fdyn)
        if "cones" in mode:
            bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
        bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0); bs.set_profiling(True)
        for _ in range(3): bs.solve()
        ms = bs.kernel_elapsed_ms(2)
        lds = 4 * (16 * (18 + (6 if "cones" in mode else 0)) * (N - 1)) / 1024
        steps = 100 * 2 * (N - 1)
        waves = (B + 15) // 16
        print(f"N={N:3d} {mode:10s} {bs.kernel_name} {ms:8.3f} ms  state LDS {lds:6.1f} KB/wave  ns per knot-step per wave-slot: "
              f"{ms * 1e6 / steps / (waves / 256):7.1f} (x waves per CU)")
        bs.close()
