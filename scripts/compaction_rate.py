#!/usr/bin/env python3
"""Tolerance-terminated solves of a big batch (second variant): single launch vs chunks with the
unconverged instances compacted in between (tinympc_set_compaction)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tinympc_julia_amd as t

for fam, B in (("quadrotor", 131072), ("cartpole", 262144), ("rocket", 32768)):
    if fam == "quadrotor":
        prob, x0 = t.problems.quadrotor(30, u_bound=0.5), t.problems.quadrotor_x0(B, seed=3)
    elif fam == "cartpole":
        prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B, seed=3)
    else:
        prob, x0 = t.problems.rocket(50), t.problems.rocket_x0(B, seed=3)
    for chunk in (0, 10, 20, 30):
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=10)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if fam == "rocket":
            xr, ur = t.problems.rocket_refs(50); bs.set_x_ref(xr); bs.set_u_ref(ur)
        bs.set_warm_start(False); bs.set_compaction(chunk); bs.set_x0(x0)
        for _ in range(2): bs.solve()
        t0 = time.perf_counter(); n = 5
        for _ in range(n): bs.solve()
        dt = (time.perf_counter() - t0) / n
        st = bs.get_status()
        print(f"{fam} B={B} {bs.kernel_name} chunk={chunk:2d}: {1e3*dt:7.2f} ms/solve  {B/dt:.3e} solves/s  "
              f"iters mean {st['iter'].mean():.1f} max {st['iter'].max()} solved {st['solved'].mean():.3f}", flush=True)
        bs.close()
