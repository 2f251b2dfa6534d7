"""kernel time of the rocket shape over the horizon, box-only and with cones + affine term, one-shot and with the workspace
kept, on the transposed-sets kernel and on what ran these solves before it (batch 32 768, 100 fixed iterations)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import tinympc_julia_amd as t
B = int(os.environ.get("B", 32768))
for N in [int(a) for a in os.environ.get("NS", "10,20,30,50").split(",")]:
    for mode in ("box", "fdyn+cones"):
        for warm in (False, True):
            prob = t.problems.rocket(N); x0 = t.problems.rocket_x0(B, seed=2); xr, ur = t.problems.rocket_refs(N)
            bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
            bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
            bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            if mode != "box":
                bs.set_fdyn(prob.fdyn); bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
            bs.set_warm_start(warm); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0); bs.set_profiling(True)
            for _ in range(4):
                if warm: bs.reset()
                bs.solve()
            print(f"{sys.argv[1]:10s} N={N:3d} {mode:10s} {'workspace kept' if warm else 'one-shot      '} {bs.kernel_name:18s} {bs.kernel_elapsed_ms(3):8.3f} ms", flush=True)
            bs.close()
'''
for label, env in (("default", {}), ("mfmat_all", {"TINYMPC_HIP_MFMAT_ALL": "1"}), ("no_mfmat", {"TINYMPC_HIP_NO_MFMAT": "1"})):
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, "-c", code, label], env=e, check=False)
