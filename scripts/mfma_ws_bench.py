#!/usr/bin/env python3
"""Quadrotor solves that carry the workspace (warm start, chunked with compaction): matrix-core WS variant vs the quad
kernel (TINYMPC_HIP_MFMA_ONESHOT_ONLY=1), same inputs; also checks the two agree."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tinympc_julia_amd as t

def run(mode, B, env):
    if env:
        os.environ["TINYMPC_HIP_MFMA_ONESHOT_ONLY"] = "1"
    else:
        os.environ.pop("TINYMPC_HIP_MFMA_ONESHOT_ONLY", None)
    prob, x0 = t.problems.quadrotor(30, u_bound=0.5), t.problems.quadrotor_x0(B, seed=3)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    out = []
    if mode == "warm_fixed100":
        bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1)
    elif mode == "warm_mpc10":
        bs.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=10, check_termination=1, en_state_bound=1, en_input_bound=1)
    else:
        bs.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=10, en_state_bound=1, en_input_bound=1)
        bs.set_warm_start(False); bs.set_compaction(20)
    bs.set_x0(x0)
    bs.set_profiling(True)
    for _ in range(3):
        bs.solve(); out.append(bs.get_solution()["controls"].copy())
    it = bs.get_status()["iter"].copy()
    t0 = time.perf_counter(); n = 5
    for _ in range(n): bs.solve()
    st = bs.solve_status() if hasattr(bs, "solve_status") else None
    bs.get_status()
    dt = (time.perf_counter() - t0) / n
    name = bs.kernel_name
    bs.close()
    return name, dt, out, it

for mode, B in (("warm_fixed100", 65536), ("warm_mpc10", 65536), ("chunked_tol", 131072)):
    a = run(mode, B, False); b = run(mode, B, True)
    err = max(np.abs(x - y).max() for x, y in zip(a[2], b[2]))
    print(f"{mode} B={B}: {a[0]} {1e3*a[1]:.2f} ms/solve | {b[0]} {1e3*b[1]:.2f} ms/solve | max|du| {err:.2e} iters equal {np.mean(a[3]==b[3]):.4f}", flush=True)
