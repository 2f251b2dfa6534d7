"""Headline workload (cartpole (4,1,20), 100 fixed iterations, cold one-shot) on the lean kernel, the quad kernel it replaces and
any variant libraries given (scripts/lean_variants.sh), all in ONE run (same box, same clocks); then a batch sweep and the
termination-check-live / state-bound patterns.  usage: lean_time.py [variant.so ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import numpy as np, sys, os, json
sys.path.insert(0, os.getcwd())
import tinympc_julia_amd as t
from tinympc_julia_amd import tinympc as tm
lib, mode = sys.argv[1], sys.argv[2]
if lib != "-": tm.load_library(lib)
prob = t.problems.cartpole(20, u_bound=0.5)
def run(B, kw, reps=24, xb=False):
    x0 = t.problems.cartpole_x0(B, seed=0)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    xmax = prob.x_max.copy()
    if xb: xmax[0, :] = 0.45
    bs.set_bound_constraints(prob.x_min, xmax, prob.u_min, prob.u_max)
    bs.set_warm_start(False); bs.set_x0(x0); bs.set_profiling(True)
    for _ in range(reps): bs.solve()
    ms = bs.kernel_elapsed_ms(reps - 4); name = bs.last_launch_name
    sol = bs.get_solution(); bs.close()
    return ms, name, sol
fixed = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
tag = os.path.basename(lib)
if mode == "head":
    ms, name, sol = run(65536, fixed)
    print(f"{tag:34s} {name:18s} 65536 x 100 it: {ms:.4f} ms  ({65536 / ms * 1e3:.3e} solves/s)  chk {float(np.abs(sol['controls']).sum()):.6f}", flush=True)
elif mode == "clock":   # a library built with -DTMPC_LEAN_CLOCK_PROBE: the residual slots carry (core clocks, 100 MHz ticks) of each lane's solve
    for B in (20480, 65536, 131072):
        x0 = t.problems.cartpole_x0(B, seed=0)
        bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
        bs.update_settings(**fixed); bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        bs.set_warm_start(False); bs.set_x0(x0); bs.set_profiling(True)
        for _ in range(20): bs.solve()
        stt = bs.get_status(); r = stt["residuals"]; ms = bs.kernel_elapsed_ms(12); bs.close()
        print(f"    first entry -> last: stores issued {(stt['iter'].max() - r[:, 2].min()) * 1e-5:.4f} ms, stores acknowledged {(stt['solved'].max() - r[:, 2].min()) * 1e-5:.4f} ms, "
              f"status fold done {(r[:, 3].max() - r[:, 2].min()) * 1e-5:.4f} ms (kernel by events {ms:.4f} ms)")
        clk = r[:, 0] / r[:, 1] * 0.1
        e0, e1 = r[:, 2].min(), r[:, 3].max()
        wv = r[::64]                         # one lane per wavefront
        print(f"    timeline (100 MHz ticks = 10 ns): first entry -> last loop end {(e1 - e0) * 1e-5:.4f} ms; entry spread {(wv[:, 2].max() - e0) * 1e-5:.4f} ms; "
              f"loop-end spread {(e1 - wv[:, 3].min()) * 1e-5:.4f} ms; per-wave loop time min / median / max {wv[:, 1].min() * 1e-5:.4f} / {np.median(wv[:, 1]) * 1e-5:.4f} / {wv[:, 1].max() * 1e-5:.4f} ms; "
              f"per-wave cycles min / median / max {wv[:, 0].min():.0f} / {np.median(wv[:, 0]):.0f} / {wv[:, 0].max():.0f}")
        h, _ = np.histogram((wv[:, 2] - e0) * 1e-2, bins=10)   # microseconds
        print("    entry-time histogram (10 bins over the spread):", h.tolist(), "  loop-time histogram:", np.histogram(wv[:, 1] * 1e-2, bins=10)[0].tolist())
        print(f"{tag:34s} batch {B:7d}: kernel {ms:.4f} ms; in-kernel core clock median {np.median(clk):.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f}); "
              f"cycles per solve median {np.median(r[:, 0]):.0f} = {np.median(r[:, 0]) / 100 / 1063:.2f} cycles per loop instruction", flush=True)
elif mode == "sweep":
    for B in (20480, 32768, 65536, 98304, 131072, 196608, 262144, 524288, 1048576):
        ms, name, _ = run(B, fixed, reps=12)
        print(f"{tag:34s} {name:18s} batch {B:8d}: {ms:.4f} ms  {B / ms * 1e3:.3e} solves/s", flush=True)
elif mode == "big":      # two wavefronts per SIMD (256 registers) against the 512-register variant in turn (TINYMPC_HIP_LEAN_ONE), by pattern
    for B in (131072, 262144):
        for label, kw, xb in (("fixed 100", fixed, False),
                              ("check live (tol 1e-30)", dict(abs_pri_tol=1e-30, abs_dua_tol=1e-30, max_iter=100, check_termination=1), False),
                              ("finite state bound, fixed 100", fixed, True),
                              ("state bound + check live", dict(abs_pri_tol=1e-30, abs_dua_tol=1e-30, max_iter=100, check_termination=1), True)):
            ms, name, _ = run(B, kw, reps=12, xb=xb)
            print(f"LEAN_ONE={os.environ.get('TINYMPC_HIP_LEAN_ONE', '0')} {name:14s} batch {B:7d} {label:32s}: {ms:.4f} ms", flush=True)
elif mode == "patterns":
    for label, kw, xb in (("fixed 100", fixed, False),
                          ("check live (tol 1e-30)", dict(abs_pri_tol=1e-30, abs_dua_tol=1e-30, max_iter=100, check_termination=1), False),
                          ("check live every 10", dict(abs_pri_tol=1e-30, abs_dua_tol=1e-30, max_iter=100, check_termination=10), False),
                          ("tol 1e-3 (early exits)", dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1), False),
                          ("finite state bound, fixed 100", fixed, True)):
        ms, name, _ = run(65536, kw, xb=xb)
        print(f"{tag:34s} {name:18s} {label:32s}: {ms:.4f} ms", flush=True)
'''
clk = [a for a in sys.argv[1:] if "clk" in os.path.basename(a)]
for lib in clk:
    subprocess.run([sys.executable, "-c", code, lib, "clock"], cwd=ROOT, check=False)
libs = ["-"] + [a for a in sys.argv[1:] if a not in clk]
for rep in range(2):
    for lib in libs:
        subprocess.run([sys.executable, "-c", code, lib, "head"], cwd=ROOT, check=False)
    env = dict(os.environ, TINYMPC_HIP_NO_LEAN="1")
    subprocess.run([sys.executable, "-c", code, "-", "head"], cwd=ROOT, check=False, env=env)
for lib in libs:
    subprocess.run([sys.executable, "-c", code, lib, "sweep"], cwd=ROOT, check=False)
subprocess.run([sys.executable, "-c", code, "-", "patterns"], cwd=ROOT, check=False)
subprocess.run([sys.executable, "-c", code, "-", "patterns"], cwd=ROOT, check=False, env=dict(os.environ, TINYMPC_HIP_NO_LEAN="1"))
subprocess.run([sys.executable, "-c", code, "-", "big"], cwd=ROOT, check=False)
subprocess.run([sys.executable, "-c", code, "-", "big"], cwd=ROOT, check=False, env=dict(os.environ, TINYMPC_HIP_LEAN_ONE="1"))
