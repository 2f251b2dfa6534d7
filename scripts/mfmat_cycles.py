"""cycles per ADMM iteration of the transposed-sets kernel's three phases (rollout / sets / backward sweep), from a probe build
of the library:  make -C tinympc-julia_amd/csrc MFMAC_FLAGS=-DTMPC_MFMAT_PROBE  (the residual outputs then carry s_memtime
deltas — such a build is NOT a solver; rebuild without the flag afterwards)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
from scripts.mfmat_quick import make
N = int(os.environ.get("N", 50))
prob = t.problems.rocket(N)
for B in [int(a) for a in os.environ.get("BS", "4096,32768").split(",")]:
    x0 = t.problems.rocket_x0(B, seed=2)
    for label, kw in (("fixed100", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)),
                      ("tol live", dict(abs_pri_tol=1e-30, abs_dua_tol=1e-30, max_iter=100, check_termination=1))):
        bs = make(prob, B, kw, N, False)
        bs.set_x0(x0); bs.set_profiling(True)
        for _ in range(3):
            bs.solve()
        r = bs.get_status()["residuals"]
        print(f"B={B:6d} {label:9s} {bs.kernel_name} {bs.kernel_elapsed_ms(2):7.3f} ms  cycles/iteration: rollout {r[:,0].mean():8.0f}  sets {r[:,1].mean():8.0f}  "
              f"backward {r[:,2].mean():8.0f}  (per knot: {r[:,0].mean()/(N-1):6.0f} {r[:,1].mean()/N:6.0f} {r[:,2].mean()/(N-1):6.0f})", flush=True)
        bs.close()
