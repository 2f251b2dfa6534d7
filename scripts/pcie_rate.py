"""PCIe-inclusive rate of configs[1]: host fp64 buffers in (set_x0), solve, host fp64 buffers out
(get_states + get_controls into preallocated arrays, i.e. what a Julia caller reusing its buffers pays)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tinympc_julia_amd as t
B = 65536
prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B, seed=0)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
bs.set_warm_start(False)
xs, us = np.zeros(4 * 20 * B), np.zeros(19 * B)
dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
def once():
    bs.lib.tinympc_set_x0(bs.h, dp(x0), B)
    bs.lib.tinympc_solve(bs.h)
    bs.lib.tinympc_get_states(bs.h, dp(xs))
    bs.lib.tinympc_get_controls(bs.h, dp(us))
for _ in range(3): once()
t0 = time.perf_counter(); n = 10
for _ in range(n): once()
dt = (time.perf_counter() - t0) / n
print(f"PCIe-inclusive (fp64 host buffers in/out through the C-ABI): {dt*1e3:.2f} ms per 65536-solve batch = {B/dt:.3e} solves/s")
# the same round trip through the fp32 entry points (Float32 host arrays: plain copies into buffers the caller page-locked with pin_host)
x0f = np.asfortranarray(x0.astype(np.float32))
xsf, usf = np.zeros(4 * 20 * B, dtype=np.float32), np.zeros(19 * B, dtype=np.float32)
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
for a in (x0f, xsf, usf): bs.pin_host(a)
def once32():
    bs.lib.tinympc_set_x0_f32(bs.h, fp(x0f), B)
    bs.lib.tinympc_solve(bs.h)
    bs.lib.tinympc_get_states_f32(bs.h, fp(xsf))
    bs.lib.tinympc_get_controls_f32(bs.h, fp(usf))
for _ in range(3): once32()
t0 = time.perf_counter()
for _ in range(n): once32()
dt32 = (time.perf_counter() - t0) / n
parts = []
for f in (lambda: bs.lib.tinympc_set_x0_f32(bs.h, fp(x0f), B), lambda: bs.lib.tinympc_solve(bs.h),
          lambda: bs.lib.tinympc_get_states_f32(bs.h, fp(xsf)), lambda: bs.lib.tinympc_get_controls_f32(bs.h, fp(usf))):
    t0 = time.perf_counter()
    for _ in range(n): f()
    parts.append((time.perf_counter() - t0) / n * 1e3)
assert np.abs(xsf.astype(np.float64) - xs).max() == 0.0 and np.abs(usf.astype(np.float64) - us).max() == 0.0
print(f"PCIe-inclusive (fp32 host buffers): {dt32*1e3:.2f} ms per 65536-solve batch = {B/dt32:.3e} solves/s   "
      f"(set_x0 {parts[0]:.2f}, solve {parts[1]:.2f}, get_states {parts[2]:.2f}, get_controls {parts[3]:.2f} ms)")
