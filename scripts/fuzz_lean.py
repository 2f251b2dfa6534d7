"""Randomised parity sweep of the lean kernel (admm_lean.hip.h) against the fp64 oracle: random (4, 1) families — stable or
mildly unstable A, random B, Q, R, rho — at the instantiated horizons, random input bounds (constant or per knot), finite state
bounds on random rows or none, zero or shared references, fixed-iteration / tolerance-terminated settings with random check
intervals, ragged batches of one lane per instance.  Every instance by solution at 1e-5 (tests/util.parity_every_instance).
PRECISION=2: the same sweep on the kernel's fp64-state form (precision 2; specialised on request, csrc/jit.cpp), x0 / bounds /
references rounded to fp32 on both sides, every instance at 1e-6 and iteration counts exactly.
SHAPE=nx,nu and / or HORIZONS=n1,n2,...: other shapes / horizons than the built-in (4,1,{5,10,15,20}) — the variants specialised
at the launch that needs them (csrc/jit.cpp: jit_lean_for); per-knot state bounds are then constant (one row pair).
Usage: python scripts/fuzz_lean.py [first_seed] [n_cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
from oracle import cpu_oracle
from tests.util import parity_every_instance, FP32_TOL
P2 = os.environ.get("PRECISION") == "2"
SHAPE = tuple(int(v) for v in os.environ.get("SHAPE", "4,1").split(","))
HORIZONS = [int(v) for v in os.environ.get("HORIZONS", "5,10,15,20").split(",")]
if P2 or "SHAPE" in os.environ or "HORIZONS" in os.environ:
    os.environ.pop("TINYMPC_HIP_NO_JIT", None)
    os.environ.setdefault("TINYMPC_HIP_CACHE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "jit_cache"))
f32 = lambda a: np.asfortranarray(np.asarray(a, dtype=np.float32).astype(np.float64))


def one(seed):
    rng = np.random.default_rng(seed)
    nx, nu = SHAPE
    N = int(rng.choice(HORIZONS))
    B = int(20480 + rng.integers(0, 6000))
    A = np.eye(nx) + 0.25 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= rng.uniform(0.9, 1.03) / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 10.0, nx)),
                              np.diag(rng.uniform(0.3, 3.0, nu)), float(rng.uniform(0.3, 3.0)), N)
    prob.x_min, prob.x_max = np.full((nx, N), -1e17), np.full((nx, N), 1e17)
    um = float(rng.uniform(0.2, 0.8))
    prob.u_min, prob.u_max = np.full((nu, N - 1), -um), np.full((nu, N - 1), um)
    if rng.random() < 0.3:
        prob.u_max[:, ::2] += 0.1                               # per-knot input bounds
    xb = rng.random() < 0.5
    if xb:
        for r in rng.choice(nx, size=int(rng.integers(1, 3)), replace=False):
            w = float(rng.uniform(0.15, 0.6))
            prob.x_min[r, :], prob.x_max[r, :] = -w, w
            if rng.random() < 0.4:
                prob.x_max[r, N // 2:] = 0.7 * w                # per-knot state bounds
    refs = rng.random() < 0.5
    xr = np.asfortranarray(0.15 * rng.standard_normal((nx, N))) if refs else None
    ur = np.asfortranarray(0.1 * rng.standard_normal((nu, N - 1))) if refs else None
    kw = [dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=int(rng.integers(1, 120)), check_termination=int(rng.choice([0, 1, 3, 7]))),
          dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=int(rng.integers(20, 120)), check_termination=int(rng.choice([1, 2, 5, 10])))][int(rng.integers(0, 2))]
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    if P2:                                                      # (the library takes x0, bounds and references as fp32 arrays)
        x0, xr, ur = f32(x0), (f32(xr) if refs else None), (f32(ur) if refs else None)
        fb = lambda a: np.where(np.abs(a) >= 1e16, a, f32(a))     # (+-1e17 = "no bound" stays what it is: rounded to fp32 it would read as a finite bound)
        prob.x_min, prob.x_max, prob.u_min, prob.u_max = fb(prob.x_min), fb(prob.x_max), fb(prob.u_min), fb(prob.u_max)
    tag = f"seed {seed} N={N} B={B} xb={xb} refs={refs} {kw}"

    def mk(b=None):
        o = cpu_oracle.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if xr is not None: o.set_x_ref(xr); o.set_u_ref(ur)
        return o
    ok = True
    try:
        ref = cpu_oracle.solve_batch("orc64", prob, x0, xref=xr, uref=ur, nthreads=len(os.sched_getaffinity(0)), **kw)
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(**kw)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if xr is not None: bs.set_x_ref(xr); bs.set_u_ref(ur)
        if P2: bs.set_precision(2)
        bs.set_warm_start(False); bs.set_x0(x0); bs.solve()
        name = bs.last_launch_name
        if P2 and (4 if xb else 2) * N * nx + 6 * N * nu + 50 > 450:   # (jit.cpp's register rule: fp64 slack AND dual of every state row with a state bound)
            assert name == "generic<f64>", name
        else:
            assert name == (f"lean<{nx},{nu},{N};f64>" if P2 else f"lean<{nx},{nu},{N}>"), name
        if P2:
            from tests.util import nrel_batch
            sol, st = bs.get_solution(), bs.get_status()
            assert np.array_equal(st["iter"], ref["iter"]) and np.array_equal(st["solved"], ref["solved"]), "iteration counts differ"
            ex, eu = nrel_batch(sol["states"], ref["x"]).max(), nrel_batch(sol["controls"], ref["u"]).max()
            assert ex <= 1e-6 and eu <= 1e-6, f"x {ex:.2e} u {eu:.2e}"
        else:
            parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, tol=FP32_TOL, min_same=0.9, tag=tag)
        bs.close()
    except AssertionError as e:
        ok = False
        print("FAIL", tag, str(e)[:300], flush=True)
        if P2:
            return (N, xb, refs, kw["abs_pri_tol"] > 0), ok
        # the same case on the quad kernel this calling pattern ran on before (fp32 state slack): is the miss the family's or the kernel's?
        os.environ["TINYMPC_HIP_NO_LEAN"] = "1"
        try:
            bq = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
            bq.update_settings(**kw)
            bq.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            if xr is not None: bq.set_x_ref(xr); bq.set_u_ref(ur)
            bq.set_warm_start(False); bq.set_x0(x0); bq.solve()
            try:
                parity_every_instance(bq.get_solution(), bq.get_status(), ref, mk, x0, kw, prob.rho, tol=FP32_TOL, min_same=0.9, tag="quad")
                print("     ->", bq.last_launch_name, "holds 1e-5 on this case", flush=True)
            except AssertionError as e2:
                print("     ->", bq.last_launch_name, "on the same case:", str(e2)[:160], flush=True)
            bq.close()
        finally:
            del os.environ["TINYMPC_HIP_NO_LEAN"]
    return (N, xb, refs, kw["abs_pri_tol"] > 0), ok


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    if not os.path.isfile(cpu_oracle.PORT_LIB): cpu_oracle.build(port=True, ref=False)
    tally = {}
    for seed in range(first, first + n):
        key, ok = one(seed)
        tally.setdefault(key, [0, 0])[0 if ok else 1] += 1
        if (seed - first) % 10 == 9: print("...", seed - first + 1, "cases", flush=True)
    print("(N, state bounds, shared refs, tolerance-terminated): ok / FAIL")
    for k, v in sorted(tally.items()): print(k, v[0], v[1])
    print("total", sum(v[0] for v in tally.values()), "ok,", sum(v[1] for v in tally.values()), "FAIL")
