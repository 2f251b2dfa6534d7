"""precision 2 (the reference's fp64 arithmetic end to end) on the headline workload — cartpole (4,1,20), 65 536 instances, 100
iterations, cold one-shot: the generic kernel's fp64-state form (TINYMPC_HIP_NO_JIT=1) against the lean kernel's fp64-state
variant specialised on request, with the library's default precision beside them."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import tinympc_julia_amd as t
B = 65536
prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B, 0)
for prec, kw in ((0, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)), (2, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)),
                 (2, dict(abs_pri_tol=1e-30, abs_dua_tol=1e-30, max_iter=100, check_termination=1))):
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_precision(prec); bs.set_warm_start(False); bs.set_x0(x0); bs.set_profiling(True)
    for _ in range(6): bs.solve()
    print(f"NO_JIT={os.environ.get('TINYMPC_HIP_NO_JIT', '0')}  precision {prec}  {'check live' if kw['abs_pri_tol'] > 0 else 'fixed     '}  {bs.last_launch_name:18s} {bs.kernel_elapsed_ms(3):9.3f} ms", flush=True)
    bs.close()
'''
env = dict(os.environ, TINYMPC_HIP_CACHE=os.path.join(ROOT, "gpurun_out", "jit_cache"))
env.pop("TINYMPC_HIP_NO_JIT", None)
subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env)
subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, TINYMPC_HIP_NO_JIT="1"))
