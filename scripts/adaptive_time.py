"""adaptive-rho solve of 65 536 cartpoles (PROB=quadrotor N=30: the shape the reference's adaptive rho is built for): kernel, time, and
agreement of the fast ADP variants (quad / matrix-core by default, then the stream kernel's) with the generic kernel"""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
B = int(os.environ.get("B", 65536))
which = os.environ.get("PROB", "cartpole")
NH = int(os.environ.get("N", 10 if which == "cartpole" else 20))
tol = float(os.environ.get("TOL", 1e-3))
prob = t.problems.cartpole(NH, u_bound=0.5) if which == "cartpole" else t.problems.quadrotor(NH, u_bound=0.5)
x0 = t.problems.cartpole_x0(B, seed=3) if which == "cartpole" else t.problems.quadrotor_x0(B, seed=3)
outs = []
for env in ((), ("TINYMPC_HIP_NO_QUAD_ADP", "TINYMPC_HIP_NO_MFMA_ADP"), ("TINYMPC_HIP_NO_QUAD_ADP", "TINYMPC_HIP_NO_MFMA_ADP", "TINYMPC_HIP_NO_STREAM_ADP")):
    for v in ("TINYMPC_HIP_NO_QUAD_ADP", "TINYMPC_HIP_NO_STREAM_ADP", "TINYMPC_HIP_NO_MFMA_ADP"):
        os.environ.pop(v, None)
    for v in env:
        os.environ[v] = "1"
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(abs_pri_tol=tol, abs_dua_tol=tol, max_iter=100, check_termination=1)
    bs.set_adaptive_rho(True)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_warm_start(False); bs.set_x0(x0); bs.set_profiling(True)
    for _ in range(3):
        bs.reset()
        bs.solve()
    ms = bs.kernel_elapsed_ms(2)
    sol, st = bs.get_solution(), bs.get_status()
    rho = bs.get_adaptive_state()['rho']
    outs.append((bs.kernel_name, ms, sol, st, rho))
    print(f"{bs.kernel_name:16s} {ms:8.3f} ms  iters mean {st['iter'].mean():.1f}  solved {st['solved'].mean():.3f}  rho range {(float(np.min(rho)), float(np.max(rho)))}")
    bs.close()
b = outs[-1]
for a in outs[:-1]:
    same = a[3]["iter"] == b[3]["iter"]
    print(a[0], "vs", b[0], ": same iteration count:", same.mean(), " max |du| on those:",
          np.abs(a[2]["controls"] - b[2]["controls"])[..., same].max() if same.any() else None,
          " max |drho|:", np.abs(np.asarray(a[4]) - np.asarray(b[4]))[same].max())
