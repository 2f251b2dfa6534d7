"""kernel time of config 4 (rocket, cones + affine term) over batch size / calling pattern on the transposed-sets kernel"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_julia_amd as t
from scripts.mfmat_quick import make
N = int(os.environ.get("N", 50))
prob = t.problems.rocket(N)
for B in [int(a) for a in os.environ.get("BS", "4096,16384,32768").split(",")]:
    x0 = t.problems.rocket_x0(B, seed=2)
    for label, kw, warm in (("fixed100 one-shot", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1), False),
                            ("fixed100 warm", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1), True),
                            ("tol live one-shot", dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1), False)):
        if os.environ.get("ONLY") and os.environ["ONLY"] not in label:
            continue
        bs = make(prob, B, kw, N, warm)
        bs.set_x0(x0); bs.set_profiling(True)
        for _ in range(5):
            if warm:
                bs.reset()
            bs.solve()
        print(f"B={B:6d} {label:20s} {bs.kernel_name} {bs.kernel_elapsed_ms(3):8.3f} ms  iters {bs.get_status()['iter'].mean():.1f}", flush=True)
        bs.close()
