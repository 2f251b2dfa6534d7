"""first contact of the transposed-sets kernel (mfmat): bitwise against the three-wavefront kernel on one-shot solves,
warm-started sequences against the fp64 oracle, timing of config 4 in its three calling patterns"""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
from oracle import cpu_oracle

CONES = ([0], [3], [0.25], [0], [3], [0.5])


def make(prob, B, kw, N, warm, cones=True, fdyn=True):
    xr, ur = t.problems.rocket_refs(N)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if fdyn:
        bs.set_fdyn(prob.fdyn)
    if cones:
        bs.set_cone_constraints(*CONES)
    bs.set_warm_start(warm)
    bs.set_x_ref(xr)
    bs.set_u_ref(ur)
    return bs


def one(N, B, kw, env):
    code = f"""
import os, sys, numpy as np
sys.path.insert(0, {os.getcwd()!r})
import tinympc_julia_amd as t
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from mfmat_quick import make
prob = t.problems.rocket({N}); x0 = t.problems.rocket_x0({B}, seed=2)
bs = make(prob, {B}, {kw!r}, {N}, False)
bs.set_x0(x0); bs.solve()
sol = bs.get_solution(); st = bs.get_status()
np.savez('/tmp/mq_out.npz', x=sol['states'], u=sol['controls'], it=st['iter'], res=st['residuals'], name=bs.kernel_name)
"""
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, "-c", code], check=True, env=e)
    return dict(np.load("/tmp/mq_out.npz"))


if __name__ == "__main__":
    what = sys.argv[1:] or ["bit", "warm", "time"]
    if "bit" in what:
        for N in (50, 10):
            for kw in (dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=1),
                       dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1),
                       dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=95, check_termination=10)):
                a = one(N, 37, kw, {})
                b = one(N, 37, kw, {"TINYMPC_HIP_NO_MFMAT": "1"})
                print(N, kw["max_iter"], kw["check_termination"], a["name"], b["name"], "bit-equal x,u:", np.array_equal(a["x"], b["x"]),
                      np.array_equal(a["u"], b["u"]), "iter equal:", np.array_equal(a["it"], b["it"]), "max |du|:", np.abs(a["u"] - b["u"]).max(),
                      "res equal:", np.array_equal(a["res"], b["res"]), flush=True)
    if "warm" in what:
        cpu_oracle.build(port=True, ref=False)
        for N, kw in ((10, dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)),
                      (10, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=7, check_termination=1)),
                      (50, dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=40, check_termination=1)),
                      (50, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=15, check_termination=5))):
            B = 21
            prob = t.problems.rocket(N)
            x0 = t.problems.rocket_x0(B, seed=5)
            xr, ur = t.problems.rocket_refs(N)
            bs = make(prob, B, kw, N, True)
            orcs = []
            for b in range(B):
                o = cpu_oracle.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
                o.update_settings(**kw)
                o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
                o.set_fdyn(prob.fdyn); o.set_cone_constraints(*CONES); o.set_x_ref(xr); o.set_u_ref(ur)
                orcs.append(o)
            x = x0.copy()
            for step in range(4):
                bs.set_x0(x); bs.solve()
                sol, st, ws = bs.get_solution(), bs.get_status(), bs.get_workspace()
                eu = ex = ed = ev = 0.0
                same = 0
                for b in range(B):
                    o = orcs[b]
                    o.set_x0(x[:, b]); o.solve(); r = o.get_solution(); w = o.get_state()
                    if r["iter"] == st["iter"][b]:
                        same += 1
                        eu = max(eu, np.abs(sol["controls"][:, :, b] - r["u"]).max() / np.abs(r["u"]).max())
                        ex = max(ex, np.abs(sol["states"][:, :, b] - r["x"]).max() / np.abs(r["x"]).max())
                        ed = max(ed, np.abs(ws["d"][:, :, b] - w["d"]).max() / max(1e-30, np.abs(w["d"]).max()))
                        ev = max(ev, np.abs(ws["v"][:, :, b] - w["v"]).max() / max(1e-30, np.abs(w["v"]).max()))
                print(f"N={N} {kw['max_iter']}/{kw['check_termination']} step {step} {bs.kernel_name}: same iter {same}/{B} "
                      f"(gpu iters {st['iter'].min()}..{st['iter'].max()}) rel err u {eu:.2e} x {ex:.2e} ws d {ed:.2e} ws v {ev:.2e}", flush=True)
                x = prob.A @ x + prob.B @ sol["controls"][:, 0, :] + prob.fdyn[:, None]
            bs.close()
    if "time" in what:
        B, N = 32768, 50
        prob = t.problems.rocket(N); x0 = t.problems.rocket_x0(B, seed=2)
        for label, kw, warm, env in (
                ("fixed100 one-shot mfmat", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1), False, {}),
                ("fixed100 warm mfmat", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1), True, {}),
                ("tol live one-shot mfmat", dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1), False, {}),
                ("tol live warm mfmat", dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1), True, {})):
            bs = make(prob, B, kw, N, warm)
            bs.set_x0(x0); bs.set_profiling(True)
            for _ in range(6):
                if warm:
                    bs.reset()
                bs.solve()
            print(f"{label:28s} {bs.kernel_name} {bs.kernel_elapsed_ms(4):8.3f} ms  iters {bs.get_status()['iter'].mean():.1f}", flush=True)
            bs.close()
