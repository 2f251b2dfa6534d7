"""where and when the tiles of a config-4 launch ran (probe build, TINYMPC_HIP_MFMAC_DEBUG=32): tiles per CU over time"""
import numpy as np, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TINYMPC_HIP_MFMAC_DEBUG"] = "32"
import tinympc_julia_amd as t
B, N = int(os.environ.get("B", 32768)), 50
prob = t.problems.rocket(N); x0 = t.problems.rocket_x0(B, seed=2); xr, ur = t.problems.rocket_refs(N)
bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
bs.set_fdyn(prob.fdyn); bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0); bs.set_profiling(True)
for _ in range(3): bs.solve()
ms = bs.kernel_elapsed_ms(2)
r = bs.get_status()["residuals"][::16].astype(np.float64)
start = r[:, 0] + 65536.0 * r[:, 1]; dur = r[:, 2]; cu = r[:, 3].astype(int)
start -= start.min()
end = start + dur
print(f"{bs.kernel_name} {ms:.3f} ms; tiles {len(r)}; distinct CUs {len(set(cu))}; span {end.max()/100:.1f} us (100 MHz clock)")
print("tile duration us: min %.0f median %.0f max %.0f" % (dur.min()/100, np.median(dur)/100, dur.max()/100))
per_cu = collections.Counter(cu)
print("tiles per CU: min %d max %d; histogram %s" % (min(per_cu.values()), max(per_cu.values()), sorted(collections.Counter(per_cu.values()).items())))
for frac in (0.1, 0.3, 0.5, 0.7, 0.9):
    tq = frac * end.max()
    live = (start <= tq) & (end > tq)
    lc = collections.Counter(cu[live])
    print(f"t = {frac:.1f} span: {live.sum()} tiles live, per-CU live histogram {sorted(collections.Counter(lc.values()).items())}")
order = np.argsort(start)
print("start times of tiles (us) deciles:", np.round(np.percentile(start, [0, 10, 25, 50, 75, 90, 100]) / 100, 1))
c0 = cu[order][0]
print("tile order on that CU; CU", c0, "tiles (start us, dur us):", [(round(start[i]/100), round(dur[i]/100)) for i in order if cu[i] == c0])
