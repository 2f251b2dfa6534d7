"""rocket solves (32 768 instances) with what the built-in `mfmat` entries do not compile — two cones on the state side,
linear-inequality rows — on the unit specialised at setup for exactly that layout (csrc/jit.cpp) and on the stream kernel
they ran on before (TINYMPC_HIP_NO_JIT=1): one-shot with 100 fixed iterations, and tolerance-terminated with the check live
every iteration (the reference's rocket loop).  N=20 by default (N=<horizon> in the environment)."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(os.environ.get("N", "20"))

CASES = {
    "one cone per side (built-in layout, for scale)": dict(cones=([0], [3], [0.25], [0], [3], [0.5]), lin=None),
    "two state cones + input cone": dict(cones=([0], [3], [0.25], [0, 3], [3, 3], [0.5, 1.5]), lin=None),
    "two state cones + input cone + one state row": dict(cones=([0], [3], [0.25], [0, 3], [3, 3], [0.5, 1.5]), lin=([[0.0, 0.0, -1.0, 0.0, 0.0, 0.3]], [0.5], None, [])),
    "cones + rows on both sides": dict(cones=([0], [3], [0.25], [0], [3], [0.5]),
                                       lin=([[0.0, 0.0, -1.0, 0.0, 0.0, 0.3]], [0.5], [[1.0, 1.0, 0.0], [-1.0, 1.0, 0.0]], [6.0, 6.0])),
}


def run(name, env, tol):
    code = f"""
import os, sys, numpy as np
sys.path.insert(0, {os.getcwd()!r})
import tinympc_julia_amd as t
N, B = {N}, 32768
c = {CASES[name]!r}
prob = t.problems.rocket(N); xr, ur = t.problems.rocket_refs(N)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol={tol[0]}, abs_dua_tol={tol[1]}, max_iter=100, check_termination=1)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max); bs.set_fdyn(prob.fdyn)
bs.set_cone_constraints(*c['cones'])
if c['lin'] is not None:
    Ax, bx, Au, bu = c['lin']
    bs.set_linear_constraints(np.array(Ax), bx, np.zeros((0, 3)) if Au is None else np.array(Au), bu)
bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur)
bs.set_x0(t.problems.rocket_x0(B, seed=2)); bs.set_profiling(True)
for _ in range(4): bs.solve()
st = bs.get_status()
print(f"{{bs.kernel_name:34s}} {{bs.kernel_elapsed_ms(3):8.3f}} ms  mean iter {{st['iter'].mean():6.1f}}", end="")
"""
    e = dict(os.environ); e.update(env)
    cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "jit_cache")
    os.makedirs(cache, exist_ok=True)
    e.setdefault("TINYMPC_HIP_CACHE", cache)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    if r.returncode:
        return "FAILED: " + r.stderr.strip().split("\n")[-1]
    return r.stdout.strip().split("\n")[-1]


for label, tol in (("100 fixed iterations", (0.0, 0.0)), ("tolerances 2e-3 / 1e-3, check every iteration", (2e-3, 1e-3))):
    print(f"--- N = {N}, 32 768 instances, {label} ---", flush=True)
    for name in CASES:
        print(f"{name:48s} {run(name, {}, tol)}   |   {run(name, {'TINYMPC_HIP_NO_JIT': '1', 'TINYMPC_HIP_NO_MFMAT': '1', 'TINYMPC_HIP_NO_MFMAC': '1'}, tol)}", flush=True)
