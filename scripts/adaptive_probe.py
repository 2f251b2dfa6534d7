"""GPU-box probe: adaptive rho on the generic kernel against the golden G9 cases and the fp32/fp64 CPU restatements."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t  # noqa: E402
from oracle import cpu_oracle  # noqa: E402
from tests.util import cm, golden_names, load_golden, nrel, problem_of  # noqa: E402

cpu_oracle.build(ref=False)
for name in [n for n in golden_names() if n.startswith("G9")]:
    g = load_golden(name)
    prob = problem_of(g)
    B = g["batch"]
    x0 = cm(g["x0"], prob.nx, B)
    dK, dP = cm(g["dKinf_drho"], prob.nu, prob.nx), cm(g["dPinf_drho"], prob.nx, prob.nx)
    for prec in (0, 1):
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.set_precision(prec)
        bs.update_settings(**g["settings"])
        if prob.has_bounds():
            bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        bs.set_sensitivity(dK, dP)
        a = g["adaptive"]
        bs.set_adaptive_rho(True, a["rho_min"], a["rho_max"], a["clip"])
        bs.set_x0(x0)
        nsolves = len(g["expect"][0])
        for k in range(nsolves):
            bs.solve()
            sol, st, ad = bs.get_solution(), bs.get_status(), bs.get_adaptive_state()
            for b in range(B):
                e = g["expect"][b][k]
                # fp32 CPU restatement on the same case
                print(name, "prec", prec, "solve", k, "inst", b, bs.kernel_name, "iter", st["iter"][b], e["iter"],
                      "rho", ad["rho"][b], e["rho"],
                      "ex %.2e eu %.2e" % (nrel(sol["states"][:, :, b], cm(e["x"], prob.nx, prob.N)),
                                           nrel(sol["controls"][:, :, b], cm(e["u"], prob.nu, prob.N - 1))),
                      "eK %.2e" % nrel(ad["Kinf"][:, :, b], cm(e["Kinf"], prob.nu, prob.nx)))
    # how far the fp32 CPU restatement lands from the same reference outputs
    for b in range(B):
        o = cpu_oracle.CpuSolver("orc32", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
        o.update_settings(**g["settings"])
        if prob.has_bounds():
            o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        o.set_sensitivity(dK, dP)
        o.set_adaptive_rho(1, a["rho_min"], a["rho_max"], a["clip"])
        o.set_x0(x0[:, b])
        for k in range(len(g["expect"][b])):
            o.solve()
            r = o.get_solution()
            e = g["expect"][b][k]
            print(name, "orc32 solve", k, "inst", b, "iter", r["iter"], e["iter"], "rho", o.get_adapted()["rho"], e["rho"],
                  "ex %.2e eu %.2e" % (nrel(r["x"], cm(e["x"], prob.nx, prob.N)), nrel(r["u"], cm(e["u"], prob.nu, prob.N - 1))))
