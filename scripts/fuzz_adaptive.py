"""Randomised sweep of the adaptive-rho kernels (quad ADP for the cartpole shapes, the matrix-core ADP variant for the quadrotor
shape at its compiled horizons — MFMA=1 restricts the sweep to it —, stream ADP for every other instantiated (nx, nu)) against the generic kernel, which the compiled-reference fixtures G9a-d pin: same iteration counts, rho within
1e-5 relative, solutions within 1e-5."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
from tests.util import nrel_batch

def run(prob, x0, kw, clip, refs, generic):
    for v in ("TINYMPC_HIP_NO_QUAD_ADP", "TINYMPC_HIP_NO_STREAM_ADP", "TINYMPC_HIP_NO_MFMA_ADP"):
        os.environ.pop(v, None)
    if generic:
        os.environ["TINYMPC_HIP_NO_QUAD_ADP"] = os.environ["TINYMPC_HIP_NO_STREAM_ADP"] = os.environ["TINYMPC_HIP_NO_MFMA_ADP"] = "1"
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=x0.shape[1])
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_adaptive_rho(True, 0.1, 10.0, clip)
    if refs is not None:
        bs.set_x_ref(refs[0]); bs.set_u_ref(refs[1])
    bs.set_x0(x0)
    outs = []
    for _ in range(2):                       # second solve: warm workspace + the adapted cache persists
        bs.solve()
        outs.append((bs.get_solution(), bs.get_status(), bs.get_adaptive_state()))
    name = bs.kernel_name
    bs.close()
    return name, outs

names, bad = {}, 0
s0, n = int(os.environ.get("SEED0", 0)), int(os.environ.get("CASES", 60))
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(seed)
    if rng.random() < 0.4:
        nx, nu, N = 4, 1, int(rng.choice([2, 10, 20]))
    else:
        nx, nu = [(2, 1), (3, 2), (4, 2), (6, 3), (8, 2), (12, 4), (4, 1)][int(rng.integers(0, 7))]
        N = int(rng.integers(3, 26))
    if os.environ.get("MFMA"):               # the matrix-core ADP variant (round 3): the quadrotor shape at its compiled horizons
        nx, nu, N = 12, 4, int(rng.choice([10, 15, 20, 25, 30]))
    B = int(rng.integers(3, 40))
    A = np.eye(nx) + 0.2 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= rng.uniform(0.9, 0.995) / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)),
                              np.diag(rng.uniform(0.5, 3.0, nu)), float(rng.uniform(0.5, 2.0)), N)
    prob.x_min, prob.x_max = -rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N)), rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N))
    prob.u_min, prob.u_max = -rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1)), rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1))
    if rng.random() < 0.3:
        prob.x_min[:], prob.x_max[:] = -1e17, 1e17
    refs = (0.2 * rng.standard_normal((nx, N)), 0.1 * rng.standard_normal((nu, N - 1))) if rng.random() < 0.5 else None
    if refs is not None and rng.random() < 0.4:   # every instance its own references
        refs = (np.asfortranarray(refs[0][:, :, None] + 0.1 * rng.standard_normal((nx, N, B))),
                np.asfortranarray(refs[1][:, :, None] + 0.05 * rng.standard_normal((nu, N - 1, B))))
    kw = [dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=int(rng.integers(12, 60)), check_termination=int(rng.choice([1, 4]))),
          dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=int(rng.integers(20, 80)), check_termination=int(rng.choice([1, 5])))][int(rng.integers(0, 2))]
    clip = bool(rng.integers(0, 2))
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    try:
        na, a = run(prob, x0, kw, clip, refs, False)
        nb, b = run(prob, x0, kw, clip, refs, True)
    except Exception as e:
        print("ERROR", seed, nx, nu, N, str(e)[:200]); bad += 1; continue
    names[na] = names.get(na, 0) + 1
    assert nb == "generic", nb
    ok = True
    for k in range(2):
        (sa, ta, aa), (sb, tb, ab) = a[k], b[k]
        same = ta["iter"] == tb["iter"]
        ex = nrel_batch(sa["controls"], sb["controls"])[same].max() if same.any() else 0.0
        er = (np.abs(aa["rho"] - ab["rho"]) / ab["rho"])[same].max() if same.any() else 0.0
        if same.mean() < 0.9 or ex > 1e-5 or er > 1e-5:
            ok = False
            print("FAIL", seed, na, (nx, nu, N, B), kw, "solve", k, "same", same.mean(), "du", ex, "drho", er, flush=True)
    bad += 0 if ok else 1
print("cases by kernel:", names, "failures:", bad)
