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
