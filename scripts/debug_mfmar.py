import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
N, B = 50, 37
prob = t.problems.rocket(N); x0 = t.problems.rocket_x0(B, seed=2); xr, ur = t.problems.rocket_refs(N)
for mode in ("fdyn+cones", "fdyn", "cones", "none"):
  for iters in (1, 2, 3, 10, 60):
    outs = []
    for env in (None, "1"):
        if env: os.environ["TINYMPC_HIP_NO_MFMAR"] = "1"
        else: os.environ.pop("TINYMPC_HIP_NO_MFMAR", None)
        os.environ["TINYMPC_HIP_MFMAC_ALL"] = "1"
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters, check_termination=1)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if "fdyn" in mode: bs.set_fdyn(prob.fdyn)
        if "cones" in mode: bs.set_cone_constraints([0], [3], [0.25], [0], [3], [0.5])
        bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0)
        bs.solve()
        outs.append((bs.kernel_name, bs.get_solution(), bs.get_status()))
        bs.close()
    dx = np.abs(outs[0][1]["states"] - outs[1][1]["states"]); du = np.abs(outs[0][1]["controls"] - outs[1][1]["controls"])
    print(mode, iters, outs[0][0], outs[1][0], "max dx %.3e du %.3e" % (dx.max(), du.max()),
          "worst inst", int(np.argmax(dx.max(axis=(0, 1)))), "row", int(np.argmax(dx.max(axis=(1, 2)))), "knot", int(np.argmax(dx.max(axis=(0, 2)))),
          "| u: inst", int(np.argmax(du.max(axis=(0, 1)))), "row", int(np.argmax(du.max(axis=(1, 2)))), "knot", int(np.argmax(du.max(axis=(0, 2)))),
          "res diff %.2e" % np.abs(outs[0][2]["residuals"] - outs[1][2]["residuals"]).max())
