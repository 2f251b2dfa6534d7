"""CPU tests of the product's host side: the C-ABI library loads, exports every symbol the header
declares, the host-only fp64 Riccati precompute matches the reference's cache, and the entry points
fail cleanly (-1 + message) when no solver / no GPU exists.  No compute calls are made here."""
import ctypes
import re

import numpy as np
import pytest

import tinympc_julia_amd as t
from tests.util import cm, load_golden, nrel, problem_of


def _header_functions():
    src = open(t.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src)
    skip = {"defined", "tinympc_solver"}
    return sorted({n for n in names if n not in skip and not n.isupper()})


def test_library_exports_every_declared_symbol(hip_lib):
    decl = _header_functions()
    assert len(decl) >= 40
    raw = ctypes.CDLL(t.LIB_PATH)
    for name in decl:
        assert hasattr(raw, name), f"{name} declared in include/tinympc_hip.h but not exported"
    # the reference's own entry points (bindings.cpp) that are in scope
    for name in ("setup_solver", "set_x0", "set_x_ref", "set_u_ref", "solve_mpc", "get_states", "get_controls",
                 "cleanup_solver", "update_settings", "set_bound_constraints", "set_cache_terms",
                 "set_linear_constraints", "set_cone_constraints", "print_problem_data"):
        assert name in decl
    # the Python mirror binds exactly the declared set
    from tinympc_julia_amd import tinympc
    assert sorted(tinympc.SIGNATURES) == decl


def test_no_solver_errors_are_clean(hip_lib):
    """bindings.cpp convention: -1, message, no crash ("Solver not initialized")."""
    lib = hip_lib
    lib.cleanup_solver()
    assert lib.solve_mpc(0) == -1
    assert b"not initialized" in lib.tinympc_last_error()
    buf = np.zeros(8)
    r, c = ctypes.c_int(), ctypes.c_int()
    assert lib.get_states(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(r), ctypes.byref(c)) == -1
    assert lib.set_batch_size(4) == -1
    assert lib.get_batch_size() == 0
    s = t.TinyMPCSolver()
    with pytest.raises(t.TinyMPCError):
        t.solve(s)                                    # "Solver not setup" (TinyMPC.jl:144)
    with pytest.raises(t.TinyMPCError):
        t.get_solution(s)


def test_setup_without_gpu_fails_loudly(hip_lib):
    """The product path has no CPU fallback: without a usable HIP device setup must fail, not compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    prob = t.problems.cartpole(10)
    with pytest.raises(t.TinyMPCError):
        t.setup(t.TinyMPCSolver(), prob.A, prob.B, np.zeros(4), prob.Q, prob.R, 1.0, 4, 1, 10)
    assert b"no HIP device" in hip_lib.tinympc_last_error() or b"hip" in hip_lib.tinympc_last_error().lower()


@pytest.mark.parametrize("name", ["G1_cartpole_one_solve", "G6_quadrotor_box_fixed100", "G7_rocket_box_fixed100",
                                  "G3c_test_settings_pritol5"])
def test_host_riccati_matches_reference_cache(hip_lib, name):
    """setup()'s fp64 precompute (host C++, csrc/host_setup.cpp) against the cache the compiled
    reference produced (tiny_api.cpp:124-190), incl. the rho-twice quirk."""
    g = load_golden(name)
    prob = problem_of(g)
    c = t.host_precompute(prob.A, prob.B, prob.Q, prob.R, prob.rho)
    for key, (r, cc) in dict(Kinf=(prob.nu, prob.nx), Pinf=(prob.nx, prob.nx), Quu_inv=(prob.nu, prob.nu),
                             AmBKt=(prob.nx, prob.nx)).items():
        assert nrel(c[key], cm(g["cache"][key], r, cc)) <= 1e-12, key


def test_host_riccati_singular_is_reported(hip_lib):
    A = np.eye(2)
    B = np.zeros((2, 1))
    Q = np.eye(2)
    R = -2.0 * np.eye(1)  # R + 2 rho = 0 with rho = 1 and B = 0  ->  singular
    with pytest.raises(t.TinyMPCError):
        t.host_precompute(A, B, Q, R, 1.0)


def test_problem_generators_are_seeded():
    a, b = t.problems.cartpole_x0(32, seed=0), t.problems.cartpole_x0(32, seed=0)
    assert np.array_equal(a, b) and a.shape == (4, 32)
    assert np.abs(a).max(axis=1).tolist() <= [0.5, 0.2, 0.1, 0.2]
    q = t.problems.quadrotor_x0(16, seed=1)
    assert q.shape == (12, 16) and np.abs(q).max() <= 0.3
    xr, ur = t.problems.rocket_refs(50)
    assert xr.shape == (6, 50) and ur.shape == (3, 49) and np.all(ur[2] == 10.0)
