"""Full-size parity: every instance of BASELINE configs 2 and 3 (batch 65 536, 100 fixed iterations)
against the fp64 CPU oracle, for both kernel precisions, and of config 4 (rocket N = 50, cones + affine term, batch 32 768,
randomised x0: every one of the 32 768 distinct) on the on-chip kernel.  Prints the error distribution."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("TINYMPC_HIP_CACHE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "jit_cache"))
import tinympc_julia_amd as t
from oracle import cpu_oracle

def nrel_batch(a, ref):
    den = np.abs(ref).max(axis=(0, 1)); den = np.where(den == 0, 1.0, den)
    return np.abs(a - ref).max(axis=(0, 1)) / den

out = {}
cores = len(os.sched_getaffinity(0))
for fam in ("cartpole", "quadrotor"):
    B = int(os.environ.get("FULL_B", 65536))
    if fam == "cartpole":
        prob, x0 = t.problems.cartpole(20, u_bound=0.5), t.problems.cartpole_x0(B, seed=0)
    else:
        prob, x0 = t.problems.quadrotor(30), t.problems.quadrotor_x0(B, seed=1)
    x0 = np.asfortranarray(x0.astype(np.float32).astype(np.float64))   # (the library takes x0 as an fp32 array: the oracle gets the same numbers)
    t0 = time.time()
    ref = cpu_oracle.solve_batch("orc64", prob, x0, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, nthreads=cores)
    t_cpu = time.time() - t0
    for prec in (0, 1, 2):   # (2: fp64 state end to end — the lean kernel's fp64-state variant specialised on request for cartpole, generic<f64> for the quadrotor)
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        bs.set_strict_precision(True)   # precision 1 means the fp32 kernels here, also where the matrix cores would be faster
        bs.set_precision(prec); bs.set_warm_start(False); bs.set_x0(x0)
        bs.solve()
        sol = bs.get_solution()
        ex, eu = nrel_batch(sol["states"], ref["x"]), nrel_batch(sol["controls"], ref["u"])
        key = f"{fam}_{('f64rec', 'f32', 'f64state')[prec]}"
        out[key] = dict(kernel=bs.last_launch_name, batch=B, x_max=float(ex.max()), u_max=float(eu.max()),
                        x_p999=float(np.quantile(ex, 0.999)), u_p999=float(np.quantile(eu, 0.999)),
                        x_median=float(np.median(ex)), u_median=float(np.median(eu)),
                        n_over_1e5=int(((ex > 1e-5) | (eu > 1e-5)).sum()), n_over_1e6=int(((ex > 1e-6) | (eu > 1e-6)).sum()),
                        oracle_seconds=t_cpu, oracle_threads=cores)
        print(key, json.dumps(out[key]), flush=True)
        bs.close()

# ---- config 4: cones and the affine term are not in solve_batch; one CpuSolver per worker thread, instance by instance ----
def config4():
    from concurrent.futures import ThreadPoolExecutor
    B, N = int(os.environ.get("FULL_B4", 32768)), 50
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=2)
    xr, ur = t.problems.rocket_refs(N)
    cones = ([0], [3], [prob.extra["cone_mu_u"]], [0], [3], [prob.extra["cone_mu_x"]])
    kw = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
    X, U = np.zeros((6, N, B)), np.zeros((3, N - 1, B))
    workers = min(cores, int(os.environ.get("ORACLE_THREADS", 16)))

    def work(w):
        o = cpu_oracle.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**kw)
        o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        o.set_fdyn(prob.fdyn)
        o.set_cone_constraints(*cones)
        o.set_x_ref(xr); o.set_u_ref(ur)
        for b in range(w, B, workers):
            o.reset()
            o.set_x0(x0[:, b])
            o.solve()
            r = o.get_solution()
            X[:, :, b], U[:, :, b] = r["x"], r["u"]
        o.close()

    t0 = time.time()
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(work, range(workers)))
    t_cpu = time.time() - t0
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_fdyn(prob.fdyn); bs.set_cone_constraints(*cones)
    bs.set_warm_start(False); bs.set_x_ref(xr); bs.set_u_ref(ur); bs.set_x0(x0)
    bs.solve()
    sol = bs.get_solution()
    ex_, eu_ = nrel_batch(sol["states"], X), nrel_batch(sol["controls"], U)
    out["rocket_soc_f64rec"] = dict(kernel=bs.kernel_name, batch=B, x_max=float(ex_.max()), u_max=float(eu_.max()),
                                    x_p999=float(np.quantile(ex_, 0.999)), u_p999=float(np.quantile(eu_, 0.999)),
                                    x_median=float(np.median(ex_)), u_median=float(np.median(eu_)),
                                    n_over_1e5=int(((ex_ > 1e-5) | (eu_ > 1e-5)).sum()), oracle_seconds=t_cpu,
                                    oracle_threads=workers)
    print("rocket_soc_f64rec", json.dumps(out["rocket_soc_f64rec"]), flush=True)
    bs.close()


config4()
json.dump(out, open(os.path.join("gpurun_out", "full_batch_parity.json"), "w"), indent=1)
