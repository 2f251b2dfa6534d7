#!/usr/bin/env python3
"""Where the time between back-to-back solves goes: wall time per solve_async with / without the library's
per-launch profiling events and with / without per-step stream events (what bench.py records)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tinympc_julia_amd as t

dev = torch.device("cuda", 0)
prob = t.problems.cartpole(20, u_bound=0.5)
B = 65536
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, device=0)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
bs.set_warm_start(False)
bs.set_x0(t.problems.cartpole_x0(B, seed=0))
stream = torch.cuda.current_stream(dev)
for prof in (True, False):
    for step_events in (True, False):
        bs.set_profiling(prof)
        for _ in range(5):
            bs.solve_async(stream.cuda_stream)
        torch.cuda.synchronize(dev)
        K = 200
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        t0 = time.perf_counter()
        for i in range(K):
            if step_events:
                evs[i][0].record(stream)
            bs.solve_async(stream.cuda_stream)
            if step_events:
                evs[i][1].record(stream)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        print(f"library events {prof!s:5} per-step torch events {step_events!s:5}: {1e3 * dt / K:.4f} ms per solve "
              f"(host enqueue {1e3 * t_enq / K:.4f} ms)", flush=True)
bs.close()
