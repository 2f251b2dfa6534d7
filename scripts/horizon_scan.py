"""kernel time of horizons that gained an on-chip instantiation in round 3 against the run-time-horizon stream kernel they
used to run on (65 536 instances, 100 fixed iterations, cold one-shot)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import tinympc_julia_amd as t
B = 65536
for fam, N in (("quadrotor", 10), ("quadrotor", 15), ("quadrotor", 25), ("cartpole", 5), ("cartpole", 15), ("cartpole", 30)):
    prob = t.problems.quadrotor(N, u_bound=0.5) if fam == "quadrotor" else t.problems.cartpole(N, u_bound=0.5)
    x0 = t.problems.quadrotor_x0(B, seed=1) if fam == "quadrotor" else t.problems.cartpole_x0(B, seed=0)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_warm_start(False); bs.set_x0(x0); bs.set_profiling(True)
    for _ in range(5): bs.solve()
    print(f"{sys.argv[1]:8s} {fam:9s} N={N:3d} {bs.kernel_name:18s} {bs.kernel_elapsed_ms(3):8.3f} ms", flush=True)
    bs.close()
'''
for label, env in (("default", {}), ("stream", {"TINYMPC_HIP_NO_QUAD": "1"})):
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, "-c", code, label], env=e, check=False)
