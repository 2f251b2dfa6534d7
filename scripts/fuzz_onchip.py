"""Randomised parity sweep of the on-chip matrix-core kernels (mfmar / mfmac) against the fp64 oracle: random stable (6,3)
families, horizons with and without a compiled instantiation, cones on either / both sides at random rows (state cone inside
rows 0..3 -> mfmar, else mfmac), per-knot or constant bounds, zero or shared references, with / without the affine term,
fixed-iteration and tolerance-terminated settings.  Every instance is compared by solution (tests/util.parity_every_instance).
WIDE=1: the general constraint layouts — two cones on a side (state side of (6,3), both sides of (6,4)), linear-inequality rows
on either / both sides, alone and with cones, at compiled and odd horizons — which since round 4 run on the transposed-sets
kernel specialised for the layout at the first solve (csrc/jit.cpp); layouts it does not take (registers) fall to the stream
kernel and are counted under its name.  TOL=<limit> (default 1e-5)."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
from oracle import cpu_oracle
from tests.util import parity_every_instance

WIDE = bool(os.environ.get("WIDE"))
TOL = float(os.environ.get("TOL", "1e-5"))
if WIDE:
    os.environ.pop("TINYMPC_HIP_NO_JIT", None)
    os.environ.setdefault("TINYMPC_HIP_CACHE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "jit_cache"))
else:
    os.environ["TINYMPC_HIP_NO_JIT"] = "1"


def one(seed):
    rng = np.random.default_rng(seed)
    nx, nu = 6, 3
    if WIDE and rng.random() < 0.4:
        nu = 4
    N = int(rng.choice([10, 20, 30, 50, 7, 13, 26, 41]))
    B = int(rng.integers(5, 60))
    A = np.eye(nx) + 0.2 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= rng.uniform(0.9, 0.99) / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)),
                              np.diag(rng.uniform(0.5, 3.0, nu)), float(rng.uniform(0.5, 2.0)), N)
    prob.x_min, prob.x_max = -rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N)), rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N))
    prob.u_min, prob.u_max = -rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1)), rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1))
    if rng.random() < 0.3:
        prob.x_min[:, N // 2:] -= 0.3
        prob.u_max[:, ::2] += 0.1
    if rng.random() < 0.25:
        prob.x_min[:], prob.x_max[:] = -1e17, 1e17
    fdyn = 0.02 * rng.standard_normal(nx) if rng.random() < 0.7 else None
    refs = rng.random() < 0.6
    xr = 0.2 * rng.standard_normal((nx, N)) if refs else None
    ur = 0.1 * rng.standard_normal((nu, N - 1)) if refs else None
    cu = ([int(rng.integers(0, 2))], [int(rng.integers(2, 3))], [float(rng.uniform(0.3, 1.2))]) if rng.random() < 0.7 else ([], [], [])
    if cu[0] and cu[0][0] + cu[1][0] > nu: cu = ([0], [3], cu[2])
    r = rng.random()
    if r < 0.45:
        a0 = int(rng.integers(0, 2)); q = int(rng.integers(2, 5 - a0))      # inside rows 0..3
        cx = ([a0], [q], [float(rng.uniform(0.3, 1.5))])
    elif r < 0.7:
        a0 = int(rng.integers(1, 4)); q = int(rng.integers(2, 7 - a0))      # anywhere
        cx = ([a0], [q], [float(rng.uniform(0.3, 1.5))])
    else:
        cx = ([], [], [])
    lin = None
    if WIDE:
        mode = int(rng.integers(0, 4))                      # 0: two cones, 1: linear rows, 2: both, 3: two cones on both sides (nu = 4)
        if mode in (0, 2, 3):
            split = int(rng.integers(2, 5))                 # state rows [0, split) and [split, 6), each >= 2 rows
            cx = ([0, split], [split, nx - split], [float(rng.uniform(0.3, 1.5)), float(rng.uniform(0.3, 1.5))])
            if rng.random() < 0.3:                          # a gap: the second cone one row shorter
                cx = ([0, split + 1], [split, nx - split - 1], cx[2]) if nx - split - 1 >= 2 else cx
            if nu == 4 and mode == 3:
                cu = ([0, 2], [2, 2], [float(rng.uniform(0.3, 1.2)), float(rng.uniform(0.3, 1.2))])
        if mode in (1, 2):
            mx, mu = int(rng.integers(0, 4)), int(rng.integers(0, 3))
            if mx + mu == 0:
                mx = 1
            lin = (rng.standard_normal((mx, nx)), list(rng.uniform(0.1, 0.6, mx)), rng.standard_normal((mu, nu)), list(rng.uniform(0.05, 0.3, mu)))
    cones = (cu[0], cu[1], cu[2], cx[0], cx[1], cx[2]) if (cu[0] or cx[0]) else None
    if cones is None and fdyn is None:
        fdyn = 0.02 * rng.standard_normal(nx)
    kw = [dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=int(rng.integers(20, 70)), check_termination=int(rng.choice([1, 3, 7]))),
          dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=int(rng.integers(30, 90)), check_termination=int(rng.choice([1, 5, 10])))][int(rng.integers(0, 2))]
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))

    def mk(b=None):
        o = cpu_oracle.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if fdyn is not None: o.set_fdyn(fdyn)
        if cones is not None: o.set_cone_constraints(*cones)
        if lin is not None: o.set_linear_constraints(*lin)
        if xr is not None: o.set_x_ref(xr); o.set_u_ref(ur)
        return o
    X, U = np.zeros((nx, N, B)), np.zeros((nu, N - 1, B))
    it, so, res = np.zeros(B, dtype=int), np.zeros(B, dtype=int), np.zeros((B, 4))
    for b in range(B):
        o = mk(); o.set_x0(x0[:, b]); o.solve(); r_ = o.get_solution()
        X[:, :, b], U[:, :, b], it[b], so[b], res[b] = r_["x"], r_["u"], r_["iter"], r_["solved"], r_["res"]
        o.close()
    ref = dict(x=X, u=U, iter=it, solved=so, res=res)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if fdyn is not None: bs.set_fdyn(fdyn)
    if cones is not None: bs.set_cone_constraints(*cones)
    if lin is not None: bs.set_linear_constraints(*lin)
    bs.set_warm_start(False)
    if xr is not None: bs.set_x_ref(xr); bs.set_u_ref(ur)
    bs.set_x0(x0); bs.solve()
    name = bs.kernel_name.split(" ")[0]
    try:
        parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, tol=TOL, min_same=0.0, tag=f"seed {seed}")
        ok = True
    except AssertionError as e:
        ok = False
        print("FAIL", seed, name, N, B, cones, None if lin is None else (len(lin[1]), len(lin[3])), kw, str(e)[:200], flush=True)
    bs.close()
    return name, ok

names, bad = {}, 0
for seed in range(int(os.environ.get("SEED0", 0)), int(os.environ.get("SEED0", 0)) + int(os.environ.get("CASES", 60))):
    name, ok = one(seed)
    names[name] = names.get(name, 0) + 1
    bad += 0 if ok else 1
print("cases by kernel:", names, "failures:", bad)
