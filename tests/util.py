"""Shared helpers for the parity tests."""
import glob
import json
import os

import numpy as np

import tinympc_julia_amd as t

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Parity tolerance of the fp32 HIP path against the fp64 reference:
#   max|a - b| <= 1e-5 * ||ref||_inf, per trajectory (norm-relative, not elementwise).
FP32_TOL = 1e-5
# fp64 restatement vs the compiled reference snapshot.
FP64_TOL = 1e-12


def golden_names():
    return sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLDEN, "*.json")))


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def cm(lst, r, c):
    return np.asarray(lst, dtype=np.float64).reshape((r, c), order="F")


def problem_of(g):
    p = g["problem"]
    nx, nu, N = p["nx"], p["nu"], p["N"]
    prob = t.problems.Problem(p["name"], cm(p["A"], nx, nx), cm(p["B"], nx, nu), cm(p["Q"], nx, nx),
                              cm(p["R"], nu, nu), p["rho"], N)
    if "x_min" in p:
        prob.x_min, prob.x_max = cm(p["x_min"], nx, N), cm(p["x_max"], nx, N)
        prob.u_min, prob.u_max = cm(p["u_min"], nu, N - 1), cm(p["u_max"], nu, N - 1)
    return prob


def nrel(a, ref):
    """norm-relative error max|a-ref| / ||ref||_inf (per trajectory)."""
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    den = np.abs(ref).max()
    if den == 0.0:
        return float(np.abs(a).max())
    return float(np.abs(a - ref).max() / den)


def nrel_batch(a, ref):
    """per-instance norm-relative error for (rows, knots, B) arrays -> (B,)"""
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    den = np.abs(ref).max(axis=(0, 1))
    den = np.where(den == 0.0, 1.0, den)
    return np.abs(a - ref).max(axis=(0, 1)) / den


def _ratio(res, pri_tol, dua_tol):
    """how far the four residuals are from the tolerances: converged iff < 1 (admm.cpp:99-103)"""
    res = np.asarray(res, dtype=np.float64)
    return max(res[0] / pri_tol, res[2] / pri_tol, res[1] / dua_tol, res[3] / dua_tol)


def parity_every_instance(sol, st, ref, make_oracle, x0, settings, rho, xref=None, uref=None, tol=FP32_TOL,
                          min_same=0.9, tag="", tol_each=None):
    """EVERY instance is compared by solution at `tol` ("compare converged runs by solution, not by iter").

    Where the GPU's iteration count equals the oracle's the oracle's solution is the reference.  Where it differs — the
    fp32 residual fell on the other side of the tolerance than the fp64 one — two things are required instead of
    dropping the instance:
      (1) the decision was marginal: at the iteration where the two part, the ORACLE's residual-to-tolerance ratio is
          within `band` of 1, band = 2 * tol * max(1, rho) * max(1, ||solution||_inf) / min(abs_pri_tol, abs_dua_tol),
          i.e. the parity tolerance itself expressed in residual units;
      (2) the solution is right: the oracle is re-run for that instance with the GPU's termination decision imposed
          (CpuSolver.set_forced_exit: same iteration count, same converged / max_iter exit, hence the same code path
          through admm.cpp:181-205) and the GPU's x, u must match THAT within `tol`.
    ref: dict(x (nx,N,B), u (nu,N-1,B), iter (B,), res (B,4)); make_oracle(b) -> a cold, fully configured CpuSolver for
    instance b (settings, bounds, shared references, extensions; b matters for per-instance families only); per-instance references are given here as 3-D xref / uref.
    tol_each: per-instance limits (B,) for the final comparison where a test derives one (default: `tol` for all).
    Returns the fraction of instances whose iteration count agreed."""
    it_g, so_g = np.asarray(st["iter"]), np.asarray(st["solved"])
    it_r = np.asarray(ref["iter"])
    B = len(it_g)
    pt, dt = float(settings["abs_pri_tol"]), float(settings["abs_dua_tol"])
    ct = max(1, int(settings.get("check_termination", 1)))
    so_r = np.asarray(ref["solved"]) if "solved" in ref else so_g
    # (an instance that reaches max_iter with its residual within rounding of the tolerance has the same iteration count on
    # both sides and the other solved flag: the same marginal decision as stopping one check apart, handled the same way)
    mism = np.nonzero((it_g != it_r) | (so_g != so_r))[0]
    assert len(mism) <= (1.0 - min_same) * B + 1e-9, f"{tag}: {len(mism)} of {B} iteration counts differ"
    X, U = np.array(ref["x"], dtype=np.float64), np.array(ref["u"], dtype=np.float64)
    for b in mism:
        assert pt > 0 and dt > 0, f"{tag}: iteration counts differ in a fixed-iteration solve"
        assert abs(int(it_g[b]) - int(it_r[b])) <= ct, f"{tag}: instance {b} stops {it_g[b]} vs {it_r[b]}"
        o = make_oracle(int(b))
        o.set_x0(x0[:, b])
        if xref is not None and np.ndim(xref) == 3:
            o.set_x_ref(xref[:, :, b])
        if uref is not None and np.ndim(uref) == 3:
            o.set_u_ref(uref[:, :, b])
        o.set_forced_exit(int(it_g[b]) if so_g[b] else -1)
        o.solve()
        r = o.get_solution()
        assert r["iter"] == it_g[b] and r["solved"] == so_g[b]
        scale = max(1.0, np.abs(r["x"]).max(), np.abs(r["u"]).max())
        band = 2.0 * tol * max(1.0, rho) * scale / min(pt, dt)
        if it_g[b] < it_r[b] or (it_g[b] == it_r[b] and so_g[b] == 1):   # the GPU saw convergence where the oracle, at the same iteration, did not (ratio >= 1)
            ratio = _ratio(r["res"], pt, dt)
            assert 1.0 <= ratio <= 1.0 + band, f"{tag}: instance {b} left early at ratio {ratio:.4f} (band {band:.3g})"
        else:                      # the oracle converged (ratio < 1) where the GPU went on
            ratio = _ratio(ref["res"][b], pt, dt)
            assert 1.0 - band <= ratio < 1.0, f"{tag}: instance {b} went on at ratio {ratio:.4f} (band {band:.3g})"
        X[:, :, b], U[:, :, b] = r["x"], r["u"]
        o.close()
    ex, eu = nrel_batch(sol["states"], X), nrel_batch(sol["controls"], U)
    lim = np.broadcast_to(np.asarray(tol_each if tol_each is not None else tol, dtype=np.float64), ex.shape)
    wx, wu = int(np.argmax(ex / lim)), int(np.argmax(eu / lim))
    assert ex[wx] <= lim[wx], f"{tag}: x of instance {wx} off by {ex[wx]:.3e} (limit {lim[wx]:.1e}, iter {it_g[wx]})"
    assert eu[wu] <= lim[wu], f"{tag}: u of instance {wu} off by {eu[wu]:.3e} (limit {lim[wu]:.1e}, iter {it_g[wu]})"
    return 1.0 - len(mism) / B
