"""Per-knot time of shapes specialised at setup (csrc/jit.cpp) against their nearest built-in neighbours and against the
run-time-shape kernel they ran on before (TINYMPC_HIP_NO_JIT=1): 65 536 instances, 100 fixed iterations, cold one-shot."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import numpy as np, sys, os, time
sys.path.insert(0, os.getcwd())
import tinympc_julia_amd as t
def fam(nx, nu, N, seed=11):
    rng = np.random.default_rng(seed)
    A = np.eye(nx) + 0.2 * rng.standard_normal((nx, nx)) / np.sqrt(nx); A *= 0.97 / np.abs(np.linalg.eigvals(A)).max()
    p = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)), np.diag(rng.uniform(0.5, 3.0, nu)), 1.0, N)
    p.x_min, p.x_max = np.full((nx, N), -1e17), np.full((nx, N), 1e17); p.u_min, p.u_max = np.full((nu, N - 1), -0.4), np.full((nu, N - 1), 0.4)
    return p, np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, 65536)))
for label, make in (("quadrotor N=10 (built in)", lambda: (t.problems.quadrotor(10), t.problems.quadrotor_x0(65536, 1))),
                    ("quadrotor N=12", lambda: (t.problems.quadrotor(12), t.problems.quadrotor_x0(65536, 1))),
                    ("quadrotor N=15 (built in)", lambda: (t.problems.quadrotor(15), t.problems.quadrotor_x0(65536, 1))),
                    ("rocket (6,3) N=10 box (built in)", lambda: (t.problems.rocket(10), t.problems.rocket_x0(65536, 2))),
                    ("(8,2) N=25", lambda: fam(8, 2, 25)),
                    ("(8,2) N=10", lambda: fam(8, 2, 10)),
                    ("(5,2) N=18", lambda: fam(5, 2, 18)),
                    ("cartpole N=10 (built in)", lambda: (t.problems.cartpole(10, u_bound=0.5), t.problems.cartpole_x0(65536, 0))),
                    ("cartpole N=12", lambda: (t.problems.cartpole(12, u_bound=0.5), t.problems.cartpole_x0(65536, 0))),
                    ("cartpole N=15 (built in)", lambda: (t.problems.cartpole(15, u_bound=0.5), t.problems.cartpole_x0(65536, 0))),
                    ("cartpole N=30 (quad built in)", lambda: (t.problems.cartpole(30, u_bound=0.5), t.problems.cartpole_x0(65536, 0))),
                    ("(3,2) N=16", lambda: fam(3, 2, 16))):
    prob, x0 = make()
    t0 = time.perf_counter()
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=65536)
    setup = time.perf_counter() - t0
    bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_warm_start(False); bs.set_x0(x0); bs.set_profiling(True)
    for _ in range(8): bs.solve()
    ms = bs.kernel_elapsed_ms(5)
    print(f"{os.environ.get('TINYMPC_HIP_NO_JIT', '0')}  {label:34s} {bs.last_launch_name:18s} {ms:8.3f} ms  {1e3 * ms / (prob.N - 1):8.1f} us per knot   (setup {setup:.1f} s)", flush=True)
    bs.close()
'''
env = dict(os.environ, TINYMPC_HIP_CACHE=os.path.join(ROOT, "gpurun_out", "jit_cache"))
env.pop("TINYMPC_HIP_NO_JIT", None)
print("NO_JIT  shape                              kernel              100 iterations")
subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env)
subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, TINYMPC_HIP_NO_JIT="1"))
