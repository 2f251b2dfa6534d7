import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
prob = t.problems.cartpole(20, u_bound=0.5)
B = 37
x0 = t.problems.cartpole_x0(B, seed=5); x0[:, 7] *= 6.0
def cfg(bs):
    bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=1)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_x0(x0)
def run(mk, n=3, between=False):
    bs = mk(); cfg(bs); outs = []
    for i in range(n):
        bs.solve()
        outs.append((bs.get_solution()["controls"].copy(), bs.get_workspace()))
    bs.close()
    return outs
a = run(lambda: t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, device=0))
a2 = run(lambda: t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, device=0))
b = run(lambda: t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, devices=[0]))
c = run(lambda: t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B, devices=[0, 0, 0]))
for i in range(3):
    print("solve", i, "single==single", np.array_equal(a[i][0], a2[i][0]), "single==sh1", np.array_equal(a[i][0], b[i][0]),
          "single==sh3", np.array_equal(a[i][0], c[i][0]))
    d = np.argwhere(a[i][0] != b[i][0])
    print("  differing (row,knot,inst):", d[:8].tolist())
    for k in a[i][1]:
        if not np.array_equal(a[i][1][k], b[i][1][k]):
            print("  ws", k, "differs at", np.argwhere(a[i][1][k] != b[i][1][k])[:5].tolist())
