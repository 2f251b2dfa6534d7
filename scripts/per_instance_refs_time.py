"""rocket solves with PER-INSTANCE references (every instance its own trajectory to track), 32 768 instances, 100 fixed
iterations: the transposed-sets kernel (round 3: a second set of LDS cells per tile) against the stream kernel these ran on"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(N, warm, env):
    code = f"""
import os, sys, numpy as np
sys.path.insert(0, {os.getcwd()!r})
import tinympc_julia_amd as t
N, B = {N}, 32768
rng = np.random.default_rng(1)
prob = t.problems.rocket(N); xr, ur = t.problems.rocket_refs(N)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max); bs.set_fdyn(prob.fdyn)
bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
bs.set_warm_start({warm})
bs.set_x_ref(np.asfortranarray(xr[:, :, None] * (1.0 + 0.1 * rng.standard_normal((1, 1, B)))))
bs.set_u_ref(np.asfortranarray(np.repeat(ur[:, :, None], B, axis=2)))
bs.set_x0(t.problems.rocket_x0(B, seed=2)); bs.set_profiling(True)
for _ in range(4): bs.solve()
print(f"{{bs.kernel_name:14s}} {{bs.kernel_elapsed_ms(3):8.3f}} ms", end="")
"""
    e = dict(os.environ); e.update(env)
    return subprocess.run([sys.executable, "-c", code], check=True, env=e, capture_output=True, text=True).stdout.strip().split("\n")[-1]


for N in (10, 20, 30, 50):
    for warm in (False, True):
        print(f"N={N:3d} {'workspace kept' if warm else 'one-shot      '}  {run(N, warm, {})}   |   {run(N, warm, {'TINYMPC_HIP_NO_MFMAT': '1'})}", flush=True)
