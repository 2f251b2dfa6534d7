"""the reference's rocket example (examples/rocket_landing_constraints.jl: N = 10, tolerances 2e-3 / 1e-3, max_iter 100,
warm-started, references shifted every step, 90 steps) for a batch of rockets: fused into one launch, against the same
loop stepped by the host (one solve launch + set_x0 / set_x_ref round trips per step)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
B, N, steps = int(os.environ.get("B", 32768)), 10, 90
prob = t.problems.rocket(N)
x0 = t.problems.rocket_x0(B, seed=2)
xinit, xgoal = np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5]), np.zeros(6)
xs, us = np.zeros((6, N, steps)), np.zeros((3, N - 1, steps))
for k in range(1, steps + 1):
    for i in range(1, N + 1):
        xs[:, i - 1, k - 1] = xinit + (xgoal - xinit) * (i + k - 2) / 99
    us[2, :, k - 1] = 10.0
def make():
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_fdyn(prob.fdyn); bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
    bs.set_x_ref(xs[:, :, 0]); bs.set_u_ref(us[:, :, 0])
    return bs
bs = make(); bs.set_ref_sequence(xs, us)
for rep in range(3):
    bs.reset(); bs.set_x0(x0)
    t0 = time.perf_counter(); log = bs.mpc_rollout(steps); dt = time.perf_counter() - t0
print(f"fused: {bs.kernel_name} {B} rockets x {steps} steps in {dt*1e3:.1f} ms = {B*steps/dt:.3e} MPC steps/s "
      f"({dt/steps*1e3:.3f} ms per step; mean ADMM iterations per step {log['iter'].mean():.1f}, solved {log['solved'].mean():.3f})")
viol = (np.hypot(log["u"][0], log["u"][1]) > 0.25 * np.abs(log["u"][2]) + 1e-4).any(axis=0).sum()
print(f"       final altitude mean {log['x'][2, -1].mean():.3f} m, rockets with a thrust-cone violation {viol}")
bs.close()
bs = make()
x = x0.copy()
t0 = time.perf_counter()
for k in range(steps):
    bs.set_x0(x); bs.set_x_ref(xs[:, :, k]); bs.set_u_ref(us[:, :, k])
    bs.solve()
    u0 = bs.get_solution()["controls"][:, 0, :]
    x = prob.A @ x + prob.B @ u0 + prob.fdyn[:, None]
dt2 = time.perf_counter() - t0
print(f"host-stepped: {bs.kernel_name} {dt2*1e3:.1f} ms = {B*steps/dt2:.3e} MPC steps/s ({dt2/steps*1e3:.3f} ms per step)")
bs.close()
