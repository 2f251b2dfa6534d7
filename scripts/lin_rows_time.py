"""one-shot rocket solves (N = 30, 32 768 instances, 100 fixed iterations) with what only the LDS kernel takes on chip since
round 3 — two cones on the state side, linear-inequality rows — on `mfmac<6,3>` and on the stream kernel they ran on before"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = {
    "one cone per side (mfmar's case, for scale)": dict(cones=([0], [3], [0.25], [0], [3], [0.5]), lin=None),
    "two state cones + input cone": dict(cones=([0], [3], [0.25], [0, 3], [3, 3], [0.5, 1.5]), lin=None),
    "cones + one linear state row": dict(cones=([0], [3], [0.25], [0], [3], [0.5]), lin=([[0.0, 0.0, -1.0, 0.0, 0.0, 0.3]], [0.5], None, [])),
    "cones + linear rows on both sides": dict(cones=([0], [3], [0.25], [0], [3], [0.5]),
                                              lin=([[0.0, 0.0, -1.0, 0.0, 0.0, 0.3]], [0.5], [[1.0, 1.0, 0.0], [-1.0, 1.0, 0.0]], [6.0, 6.0])),
}


def run(name, env):
    code = f"""
import os, sys, numpy as np
sys.path.insert(0, {os.getcwd()!r})
import tinympc_julia_amd as t
N, B = 30, 32768
c = {CASES[name]!r}
prob = t.problems.rocket(N); xr, ur = t.problems.rocket_refs(N)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max); bs.set_fdyn(prob.fdyn)
bs.set_cone_constraints(*c['cones'])
if c['lin'] is not None:
    Ax, bx, Au, bu = c['lin']
    bs.set_linear_constraints(np.array(Ax), bx, np.zeros((0, 3)) if Au is None else np.array(Au), bu)
bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur)
bs.set_x0(t.problems.rocket_x0(B, seed=2)); bs.set_profiling(True)
for _ in range(4): bs.solve()
print(f"{{bs.kernel_name:14s}} {{bs.kernel_elapsed_ms(3):8.3f}} ms", end="")
"""
    e = dict(os.environ); e.update(env)
    return subprocess.run([sys.executable, "-c", code], check=True, env=e, capture_output=True, text=True).stdout.strip().split("\n")[-1]


for name in CASES:
    print(f"{name:46s} {run(name, {'TINYMPC_HIP_NO_MFMAT': '1', 'TINYMPC_HIP_MFMAC_WIDE': '1'})}   |   {run(name, {'TINYMPC_HIP_NO_MFMAT': '1', 'TINYMPC_HIP_NO_MFMAC': '1'})}", flush=True)
