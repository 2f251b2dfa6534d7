#!/usr/bin/env python3
"""Closed-loop steps per second (tinympc_mpc_rollout), quadrotor: matrix-core chain vs the quad kernel's fused loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tinympc_julia_amd as t

B, steps = 65536, 20
prob, x0 = t.problems.quadrotor(30, u_bound=0.5), t.problems.quadrotor_x0(B, seed=3)
logs = []
for env in (False, True):
    if env:
        os.environ["TINYMPC_HIP_MFMA_ONESHOT_ONLY"] = "1"
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=10, check_termination=1)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_x0(x0)
    log = bs.mpc_rollout(steps)
    logs.append(log)
    bs.reset(); bs.set_x0(x0)
    import ctypes
    t0 = time.perf_counter()
    bs.lib.tinympc_mpc_rollout(bs.h, steps, ctypes.c_void_p(0))   # launches + status; the logs stay on the device
    dt = time.perf_counter() - t0
    print(f"{bs.kernel_name}: {steps} steps x {B} instances in {1e3*dt:.1f} ms = {1e3*dt/steps:.2f} ms/step "
          f"({B*steps/dt:.3e} MPC steps/s), mean iters {np.abs(log['iter']).mean():.1f}", flush=True)
    bs.close()
print("max |du| between the two:", np.abs(logs[0]["u"] - logs[1]["u"]).max(), "iters equal:", np.mean(logs[0]["iter"] == logs[1]["iter"]))
