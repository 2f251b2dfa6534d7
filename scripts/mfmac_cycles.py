"""cycle counts of the mfmac kernel's phases (TINYMPC_HIP_MFMAC_DEBUG & 8: the residual outputs carry s_memtime deltas per
knot step: forward sweep, wave 0's barrier wait inside it, backward sweep, wave 1's barrier wait)"""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
B = int(os.environ.get("B", 8192))
N = int(os.environ.get("N", 50))
for dbg in [int(a) for a in os.environ.get("DBG", "8,10,12,14,9").split(",")]:
    os.environ["TINYMPC_HIP_MFMAC_DEBUG"] = str(dbg)
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=2)
    xr, ur = t.problems.rocket_refs(N)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_fdyn(prob.fdyn)
    bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
    bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0); bs.set_profiling(True)
    for _ in range(3): bs.solve()
    ms = bs.kernel_elapsed_ms(2)
    r = bs.get_status()["residuals"]
    if dbg & 16:
        rr = r[::16].astype(int)
        import collections
        print("simd of (wave0, wave1, wave2) per tile:", collections.Counter(map(tuple, rr[:, :3])).most_common(12))
        by_cu = collections.defaultdict(list)
        for row in rr[:64]: by_cu[row[3]].append(tuple(row[:3]))
        print("first tiles grouped by (cu, se):", dict(list(by_cu.items())[:8]))
        bs.close(); continue
    print(f"dbg={dbg:2d} {bs.kernel_name} {ms:7.3f} ms | per knot step: fwd {np.median(r[:,0]):7.1f} (barrier wait {np.median(r[:,1]):6.1f})  "
          f"bwd {np.median(r[:,2]):7.1f}  wave1 barrier wait {np.median(r[:,3]):6.1f}   ticks; ns/step {ms*1e6/100/(N-1)/max(1,B//8192):.1f}")
    bs.close()
