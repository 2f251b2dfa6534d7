"""Python mirror of the reference's Julia module over the C-ABI (ctypes in place of ccall).

Same names, argument meaning and error behaviour as reference src/TinyMPC.jl:
  TinyMPCSolver (:34-48), setup (:55-112), set_x0 / set_x_ref / set_u_ref (:115-141),
  solve (:143-148, returns the status without throwing), get_solution (:150-177),
  update_settings (:181-211), set_bound_constraints (:214-227), set_cache_terms (:278-292)
with a batch dimension added:
  setup(..., batch=B); set_x0 takes (nx,) or (nx, B); set_x_ref (nx, N) or (nx, N, B);
  get_solution returns states (nx, N, B) / controls (nu, N-1, B) (squeezed to 2-D for B == 1,
  i.e. exactly the reference's shapes).

Like the reference (one process-global `g_solver`, bindings.cpp:15) the module-level functions
drive ONE global solver inside libtinympc_hip.so; `BatchSolver` wraps the handle API for callers
that need several solvers or device-resident I/O.

The library is the HIP build only.  If it is missing, or no GPU is usable, calls raise — there is
no CPU fallback.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libtinympc_hip.so")

c_dp = ctypes.POINTER(ctypes.c_double)
c_fp = ctypes.POINTER(ctypes.c_float)
c_ip = ctypes.POINTER(ctypes.c_int)
c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_vp = ctypes.c_void_p


class TinyMPCError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol include/tinympc_hip.h declares
SIGNATURES = {
    "setup_solver": (c_int, [c_dp, c_int, c_int, c_dp, c_int, c_int, c_dp, c_int, c_int, c_dp, c_int,
                             c_int, c_dp, c_int, c_int, c_dbl, c_int, c_int, c_int, c_int]),
    "set_x0": (c_int, [c_dp, c_int, c_int, c_int]),
    "set_x_ref": (c_int, [c_dp, c_int, c_int, c_int]),
    "set_u_ref": (c_int, [c_dp, c_int, c_int, c_int]),
    "solve_mpc": (c_int, [c_int]),
    "get_states": (c_int, [c_dp, c_ip, c_ip]),
    "get_controls": (c_int, [c_dp, c_ip, c_ip]),
    "cleanup_solver": (None, []),
    "update_settings": (c_int, [c_dbl, c_dbl, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_dbl, c_dbl, c_int, c_int]),
    "set_bound_constraints": (c_int, [c_dp, c_int, c_int, c_dp, c_int, c_int, c_dp, c_int, c_int,
                                      c_dp, c_int, c_int, c_int]),
    "set_cache_terms": (c_int, [c_dp, c_int, c_int, c_dp, c_int, c_int, c_dp, c_int, c_int, c_dp,
                                c_int, c_int, c_int]),
    "set_sensitivity": (c_int, [c_dp, c_int, c_int, c_dp, c_int, c_int, c_dp, c_int, c_int, c_dp, c_int, c_int, c_int]),
    "get_adaptive_rho": (c_int, [c_dp, c_ip]),
    "set_warm_start": (c_int, [c_int]),
    "get_kernel_name": (ctypes.c_char_p, []),
    "print_problem_data": (c_int, [c_int]),
    "set_linear_constraints": (c_int, [c_dp, c_int, c_int, c_dp, c_int, c_dp, c_int, c_int, c_dp,
                                       c_int, c_int]),
    "set_cone_constraints": (c_int, [c_ip, c_int, c_ip, c_int, c_dp, c_int, c_ip, c_int, c_ip, c_int,
                                     c_dp, c_int, c_int]),
    "set_batch_size": (c_int, [c_int]),
    "get_batch_size": (c_int, []),
    "get_status": (c_int, [c_ip, c_ip, c_dp]),
    "reset_workspace": (c_int, []),
    "tinympc_create": (c_int, [ctypes.POINTER(c_vp), c_dp, c_dp, c_dp, c_dp, c_dbl, c_int, c_int,
                               c_int, c_int, c_int, c_int]),
    "tinympc_create_families": (c_int, [ctypes.POINTER(c_vp), c_dp, c_dp, c_dp, c_dp, c_dp, c_int, c_int, c_int,
                                        c_int, c_int, c_int]),
    "tinympc_destroy": (None, [c_vp]),
    "tinympc_update_settings": (c_int, [c_vp, c_dbl, c_dbl, c_int, c_int, c_int, c_int]),
    "tinympc_set_bound_constraints": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_set_cache_terms": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_get_cache_terms": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_set_fdyn": (c_int, [c_vp, c_dp]),
    "tinympc_set_cone_constraints": (c_int, [c_vp, c_ip, c_ip, c_dp, c_int, c_ip, c_ip, c_dp, c_int]),
    "tinympc_enable_cones": (c_int, [c_vp, c_int, c_int]),
    "tinympc_set_linear_constraints": (c_int, [c_vp, c_dp, c_int, c_dp, c_dp, c_int, c_dp]),
    "tinympc_enable_linear": (c_int, [c_vp, c_int, c_int]),
    "tinympc_set_x0": (c_int, [c_vp, c_dp, c_int]),
    "tinympc_set_x_ref": (c_int, [c_vp, c_dp, c_int]),
    "tinympc_set_u_ref": (c_int, [c_vp, c_dp, c_int]),
    "tinympc_reset": (c_int, [c_vp]),
    "tinympc_set_warm_start": (c_int, [c_vp, c_int]),
    "tinympc_solve": (c_int, [c_vp]),
    "tinympc_get_states": (c_int, [c_vp, c_dp]),
    "tinympc_get_controls": (c_int, [c_vp, c_dp]),
    "tinympc_get_status": (c_int, [c_vp, c_ip, c_ip, c_dp]),
    "tinympc_get_workspace": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_set_workspace": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_device_buffers": (c_int, [c_vp] + [ctypes.POINTER(c_vp)] * 9),
    "tinympc_set_ref_mode": (c_int, [c_vp, c_int]),
    "tinympc_solve_async": (c_int, [c_vp, c_vp]),
    "tinympc_solve_status": (c_int, [c_vp]),
    "tinympc_mpc_rollout": (c_int, [c_vp, c_int, c_vp]),
    "tinympc_get_mpc_log": (c_int, [c_vp, c_dp, c_dp, c_ip]),
    "tinympc_set_x0_f32": (c_int, [c_vp, c_fp, c_int]),
    "tinympc_get_states_f32": (c_int, [c_vp, c_fp]),
    "tinympc_get_controls_f32": (c_int, [c_vp, c_fp]),
    "tinympc_pin_host": (c_int, [c_vp, c_vp, ctypes.c_size_t]),
    "tinympc_unpin_host": (c_int, [c_vp, c_vp]),
    "set_ref_sequence": (c_int, [c_dp, c_int, c_int, c_dp, c_int, c_int, c_int]),
    "mpc_rollout": (c_int, [c_int, c_dp, c_dp, c_ip]),
    "set_x0_f32": (c_int, [c_fp, c_int, c_int, c_int]),
    "get_states_f32": (c_int, [c_fp, c_ip, c_ip]),
    "get_controls_f32": (c_int, [c_fp, c_ip, c_ip]),
    "pin_host_buffer": (c_int, [c_vp, ctypes.c_size_t]),
    "set_precision": (c_int, [c_int]),
    "unpin_host_buffer": (c_int, [c_vp]),
    "tinympc_set_ref_sequence": (c_int, [c_vp, c_dp, c_int, c_int, c_dp, c_int, c_int, c_int]),
    "tinympc_set_profiling": (c_int, [c_vp, c_int]),
    "tinympc_set_compaction": (c_int, [c_vp, c_int]),
    "tinympc_set_adaptive_rho": (c_int, [c_vp, c_int, c_dbl, c_dbl, c_int]),
    "tinympc_set_sensitivity": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_compute_sensitivity": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_get_adaptive_state": (c_int, [c_vp, c_dp, c_dp, c_dp]),
    "tinympc_kernel_elapsed_ms": (c_dbl, [c_vp]),
    "tinympc_kernel_elapsed_mean_ms": (c_dbl, [c_vp, c_int]),
    "tinympc_set_precision": (c_int, [c_vp, c_int]),
    "tinympc_kernel_name": (ctypes.c_char_p, [c_vp]),
    "tinympc_last_launch_name": (ctypes.c_char_p, [c_vp]),
    "tinympc_set_strict_precision": (c_int, [c_vp, c_int]),
    "tinympc_effective_precision": (c_int, [c_vp]),
    "tinympc_reload_switches": (c_int, [c_vp]),
    "tinympc_specialise": (c_int, [c_int, c_int, c_int, c_int]),
    "tinympc_algorithmic_bytes": (c_dbl, [c_vp]),
    "tinympc_algorithmic_flops": (c_dbl, [c_vp, c_int]),
    "tinympc_last_error": (ctypes.c_char_p, []),
    "tinympc_host_precompute": (c_int, [c_dp, c_dp, c_dp, c_dp, c_dbl, c_int, c_int, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_host_sensitivity": (c_int, [c_dp, c_dp, c_dp, c_dp, c_dbl, c_int, c_int, c_dp, c_dp, c_dp, c_dp]),
    # multi-GPU: one handle, n_gpus devices (include/tinympc_hip.h section 3)
    "tinympc_create_sharded": (c_int, [ctypes.POINTER(c_vp), c_dp, c_dp, c_dp, c_dp, c_dbl, c_int, c_int, c_int, c_int,
                                       c_int, c_ip, c_int]),
    "tinympc_sharded_destroy": (None, [c_vp]),
    "tinympc_sharded_n_shards": (c_int, [c_vp]),
    "tinympc_sharded_fold_backend": (ctypes.c_char_p, [c_vp]),
    "tinympc_sharded_shard": (c_int, [c_vp, c_int, c_ip, c_ip, c_ip, ctypes.POINTER(c_vp)]),
    "tinympc_shard_range": (None, [c_int, c_int, c_int, c_ip, c_ip]),
    "tinympc_sharded_update_settings": (c_int, [c_vp, c_dbl, c_dbl, c_int, c_int, c_int, c_int]),
    "tinympc_sharded_set_bound_constraints": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp]),
    "tinympc_sharded_set_warm_start": (c_int, [c_vp, c_int]),
    "tinympc_sharded_reset": (c_int, [c_vp]),
    "tinympc_sharded_set_precision": (c_int, [c_vp, c_int]),
    "tinympc_sharded_set_compaction": (c_int, [c_vp, c_int]),
    "tinympc_sharded_set_x0": (c_int, [c_vp, c_dp, c_int]),
    "tinympc_sharded_set_x_ref": (c_int, [c_vp, c_dp, c_int]),
    "tinympc_sharded_set_u_ref": (c_int, [c_vp, c_dp, c_int]),
    "tinympc_sharded_solve": (c_int, [c_vp]),
    "tinympc_sharded_solve_async": (c_int, [c_vp]),
    "tinympc_sharded_wait": (c_int, [c_vp]),
    "tinympc_sharded_global_status": (c_int, [c_vp, c_dp, c_ip]),
    "tinympc_sharded_get_states": (c_int, [c_vp, c_dp]),
    "tinympc_sharded_get_controls": (c_int, [c_vp, c_dp]),
    "tinympc_sharded_get_status": (c_int, [c_vp, c_ip, c_ip, c_dp]),
    "tinympc_sharded_get_workspace": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "set_gpus": (c_int, [c_int]),
    "get_gpus": (c_int, []),
}


def load_library(path=None):
    """dlopen libtinympc_hip.so and bind every declared symbol (the `_ensure_loaded` of TinyMPC.jl:14)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or _LIB_PATH
    if not os.path.isfile(path):
        raise TinyMPCError(f"TinyMPC library not found: {path} (build it with __graft_entry__.build())")
    # A process that also uses PyTorch must load torch FIRST: the torch wheel brings its own copy of the HIP / HSA runtime
    # libraries, and if this library has pulled in the system's (/opt/rocm) before, torch's later initialisation finds
    # "No HIP GPUs" (two runtimes in one process; measured on this image).  A host without
    # torch (Julia, C) is not concerned.
    import sys
    if "torch" not in sys.modules and os.environ.get("TINYMPC_HIP_NO_TORCH_PRELOAD") is None:
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _err():
    try:
        return (load_library().tinympc_last_error() or b"").decode()
    except Exception:
        return ""


def _mat(a, name=None):
    m = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if m.ndim == 1:
        m = np.asfortranarray(m.reshape(-1, 1))
    return m


def _dp(a):
    return a.ctypes.data_as(c_dp)


class TinyMPCSolver:
    """Mirror of the Julia struct (TinyMPC.jl:34-48): dims, rho, is_setup and copies of A, B, Q, R."""

    def __init__(self):
        self.nx = 0
        self.nu = 0
        self.N = 0
        self.batch = 1
        self.rho = 0.0
        self.is_setup = False
        self.A = np.zeros((0, 0))
        self.B = np.zeros((0, 0))
        self.Q = np.zeros((0, 0))
        self.R = np.zeros((0, 0))


def setup(solver, A, B, f, Q, R, rho, nx, nu, N, *, batch=1, n_gpus=1, verbose=False, abs_pri_tol=1e-3,
          abs_dua_tol=1e-3, max_iter=100, check_termination=True, adaptive_rho=False,
          adaptive_rho_min=0.1, adaptive_rho_max=10.0, adaptive_rho_clipping=True):
    """TinyMPC.jl:55-112.  Raises on failure like the Julia `error(...)`; returns the status."""
    lib = load_library()
    A, B, Q, R = _mat(A), _mat(B), _mat(Q), _mat(R)
    f = _mat(np.zeros(nx) if f is None else f)
    solver.nx, solver.nu, solver.N, solver.rho, solver.batch = nx, nu, N, float(rho), int(batch)
    solver.A, solver.B, solver.Q, solver.R = A.copy(), B.copy(), Q.copy(), R.copy()
    status = lib.setup_solver(_dp(A), A.shape[0], A.shape[1], _dp(B), B.shape[0], B.shape[1],
                              _dp(f), f.shape[0], f.shape[1], _dp(Q), Q.shape[0], Q.shape[1],
                              _dp(R), R.shape[0], R.shape[1], float(rho), nx, nu, N, 1 if verbose else 0)
    if status != 0:
        solver.is_setup = False
        raise TinyMPCError(f"Setup failed with status: {status} ({_err()})")
    if batch != 1 and lib.set_batch_size(int(batch)) != 0:
        raise TinyMPCError(f"Failed to set batch size ({_err()})")
    solver.is_setup = True
    if n_gpus != 1:
        set_gpus(solver, n_gpus)
    # TinyMPC.jl:89-104 — settings pushed with every en_* false
    update_settings(solver, abs_pri_tol=abs_pri_tol, abs_dua_tol=abs_dua_tol, max_iter=max_iter,
                    check_termination=check_termination, en_state_bound=False, en_input_bound=False,
                    en_state_soc=False, en_input_soc=False, en_state_linear=False,
                    en_input_linear=False, adaptive_rho=adaptive_rho,
                    adaptive_rho_min=adaptive_rho_min, adaptive_rho_max=adaptive_rho_max,
                    adaptive_rho_enable_clipping=adaptive_rho_clipping, verbose=verbose)
    return status


def _need_setup(solver):
    if not solver.is_setup:
        raise TinyMPCError("Solver not setup")


def set_gpus(solver, n_gpus):
    """Spread the global solver's batch over devices 0 .. n_gpus-1 of this node (contiguous shards, status all-reduced
    over RCCL): every other call then acts on the whole sharded batch.  Inputs / workspace are reset."""
    _need_setup(solver)
    if load_library().set_gpus(int(n_gpus)) != 0:
        raise TinyMPCError(f"Failed to set the number of GPUs ({_err()})")
    solver.n_gpus = int(n_gpus)
    return 0


def set_warm_start(solver, on):
    """on=False: every solve() starts from the zero workspace and keeps none (one-shot solves; the on-chip kernels
    apply); True (default): the reference's semantics, the workspace persists between solves"""
    _need_setup(solver)
    if load_library().set_warm_start(1 if on else 0) != 0:
        raise TinyMPCError(f"Failed to set warm start ({_err()})")


def kernel_name():
    """the kernel the global solver's last solve ran on"""
    return load_library().get_kernel_name().decode()


def set_batch_size(solver, batch):
    _need_setup(solver)
    if load_library().set_batch_size(int(batch)) != 0:
        raise TinyMPCError(f"Failed to set batch size ({_err()})")
    solver.batch = int(batch)
    return 0


def set_x0(solver, x0, *, verbose=False):
    """TinyMPC.jl:115-123.  x0: (nx,) broadcast to the batch, or (nx, batch)."""
    _need_setup(solver)
    m = _mat(x0)
    status = load_library().set_x0(_dp(m), m.shape[0], m.shape[1], 1 if verbose else 0)
    if status != 0:
        raise TinyMPCError(f"Failed to set initial state ({_err()})")
    return status


def _ref3(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 3:  # (rows, knots, batch) -> rows x (knots*batch), column-major
        a = np.asfortranarray(a).reshape(a.shape[0], -1, order="F")
    return _mat(a)


def set_x_ref(solver, x_ref, *, verbose=False):
    """TinyMPC.jl:125-132.  (nx, N) shared or (nx, N, batch)."""
    _need_setup(solver)
    m = _ref3(x_ref)
    status = load_library().set_x_ref(_dp(m), m.shape[0], m.shape[1], 1 if verbose else 0)
    if status != 0:
        raise TinyMPCError(f"Failed to set state reference ({_err()})")
    return status


def set_u_ref(solver, u_ref, *, verbose=False):
    """TinyMPC.jl:134-141.  (nu, N-1) shared or (nu, N-1, batch)."""
    _need_setup(solver)
    m = _ref3(u_ref)
    status = load_library().set_u_ref(_dp(m), m.shape[0], m.shape[1], 1 if verbose else 0)
    if status != 0:
        raise TinyMPCError(f"Failed to set input reference ({_err()})")
    return status


def solve(solver, *, verbose=False):
    """TinyMPC.jl:143-148 — returns the status (0 converged / 1 max_iter / -1 error), never raises on it."""
    _need_setup(solver)
    return int(load_library().solve_mpc(1 if verbose else 0))


def get_solution(solver):
    """TinyMPC.jl:150-177.  Returns dict(states=(nx,N[,B]), controls=(nu,N-1[,B]))."""
    _need_setup(solver)
    lib = load_library()
    nx, nu, N, B = solver.nx, solver.nu, solver.N, solver.batch
    sb = np.zeros(nx * N * B)
    cb = np.zeros(nu * (N - 1) * B)
    sr, sc, cr, cc = c_int(), c_int(), c_int(), c_int()
    s1 = lib.get_states(_dp(sb), ctypes.byref(sr), ctypes.byref(sc))
    s2 = lib.get_controls(_dp(cb), ctypes.byref(cr), ctypes.byref(cc))
    if s1 != 0 or s2 != 0:
        raise TinyMPCError(f"Failed to get solution ({_err()})")
    states = sb[: sr.value * sc.value].reshape((sr.value, sc.value), order="F")
    controls = cb[: cr.value * cc.value].reshape((cr.value, cc.value), order="F")
    if B > 1:
        states = states.reshape((nx, N, B), order="F")
        controls = controls.reshape((nu, N - 1, B), order="F")
    return dict(states=states, controls=controls)


def get_status(solver):
    """Batched solution->iter / solved / residuals (extension; types.hpp:32-37,128-131)."""
    _need_setup(solver)
    B = solver.batch
    it = np.zeros(B, dtype=np.int32)
    so = np.zeros(B, dtype=np.int32)
    res = np.zeros((B, 4))
    if load_library().get_status(it.ctypes.data_as(c_ip), so.ctypes.data_as(c_ip), _dp(res)) != 0:
        raise TinyMPCError(f"Failed to get status ({_err()})")
    return dict(iter=it, solved=so, residuals=res)


def set_ref_sequence(solver, x_ref_seq, u_ref_seq):
    """shared references of every step of the next closed loops, (nx, N, steps) / (nu, N-1, steps)"""
    _need_setup(solver)
    xs = np.asfortranarray(np.asarray(x_ref_seq, dtype=np.float64))
    us = np.asfortranarray(np.asarray(u_ref_seq, dtype=np.float64))
    steps = xs.shape[2]
    xs, us = xs.reshape(-1, order="F"), us.reshape(-1, order="F")
    if load_library().set_ref_sequence(_dp(xs), solver.nx, solver.N * steps, _dp(us), solver.nu, (solver.N - 1) * steps, steps) != 0:
        raise TinyMPCError(f"Failed to set the reference sequence ({_err()})")


def mpc_rollout(solver, steps):
    """`steps` fused closed-loop steps on the global solver: dict(status, x (nx, steps, B), u (nu, steps, B), iter, solved)"""
    _need_setup(solver)
    nx, nu, B = solver.nx, solver.nu, solver.batch
    x, u = np.zeros(nx * steps * B), np.zeros(nu * steps * B)
    it = np.zeros(steps * B, dtype=np.int32)
    st = int(load_library().mpc_rollout(int(steps), _dp(x), _dp(u), it.ctypes.data_as(c_ip)))
    if st < 0:
        raise TinyMPCError(f"mpc_rollout failed ({_err()})")
    it = it.reshape((steps, B), order="F")
    return dict(status=st, x=x.reshape((nx, steps, B), order="F"), u=u.reshape((nu, steps, B), order="F"),
                iter=np.abs(it), solved=(it > 0).astype(np.int32))


def reset_workspace(solver):
    _need_setup(solver)
    if load_library().reset_workspace() != 0:
        raise TinyMPCError(f"Failed to reset workspace ({_err()})")
    return 0


def update_settings(solver, *, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100,
                    check_termination=True, en_state_bound=False, en_input_bound=False,
                    en_state_soc=False, en_input_soc=False, en_state_linear=False,
                    en_input_linear=False, adaptive_rho=False, adaptive_rho_min=0.1,
                    adaptive_rho_max=10.0, adaptive_rho_enable_clipping=True, verbose=False):
    """TinyMPC.jl:181-211 — every field is sent with its keyword default, like the reference
    (so calling it later resets en_*_bound to false and the tolerances to 1e-3).
    `check_termination` may also be an int interval (extension); True/False map to 1/0."""
    ct = int(check_termination) if not isinstance(check_termination, bool) else (1 if check_termination else 0)
    status = load_library().update_settings(
        float(abs_pri_tol), float(abs_dua_tol), int(max_iter), ct, int(bool(en_state_bound)),
        int(bool(en_input_bound)), int(bool(en_state_soc)), int(bool(en_input_soc)),
        int(bool(en_state_linear)), int(bool(en_input_linear)), int(bool(adaptive_rho)),
        float(adaptive_rho_min), float(adaptive_rho_max), int(bool(adaptive_rho_enable_clipping)),
        1 if verbose else 0)
    if status != 0:
        raise TinyMPCError(f"Failed to update settings ({_err()})")
    return status


def set_bound_constraints(solver, x_min, x_max, u_min, u_max, *, verbose=False):
    """TinyMPC.jl:214-227; both bound flags are auto-enabled in the library (bindings.cpp:400-404)."""
    ms = [_mat(m) for m in (x_min, x_max, u_min, u_max)]
    args = []
    for m in ms:
        args += [_dp(m), m.shape[0], m.shape[1]]
    status = load_library().set_bound_constraints(*args, 1 if verbose else 0)
    if status != 0:
        raise TinyMPCError(f"Failed to set bound constraints ({_err()})")
    return status


def _lin_block(A, b, n):
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(-1))
    A = np.asarray(A, dtype=np.float64)
    A = _mat(A.reshape(len(b), n) if len(b) else np.zeros((0, n)))
    return A, b


def set_linear_constraints(solver, Alin_x, blin_x, Alin_u, blin_u, *, verbose=False):
    """TinyMPC.jl:229-243: Alin_x x <= blin_x, Alin_u u <= blin_u at every knot; parity unpinned."""
    _need_setup(solver)
    Ax, bx = _lin_block(Alin_x, blin_x, solver.nx)
    Au, bu = _lin_block(Alin_u, blin_u, solver.nu)
    status = load_library().set_linear_constraints(_dp(Ax), Ax.shape[0], Ax.shape[1], _dp(bx), len(bx),
                                                   _dp(Au), Au.shape[0], Au.shape[1], _dp(bu), len(bu),
                                                   1 if verbose else 0)
    if status != 0:
        raise TinyMPCError(f"Failed to set linear constraints ({_err()})")
    return status


def set_equality_constraints(solver, Aeq_x, beq_x, Aeq_u=None, beq_u=None):
    """TinyMPC.jl:261-270: equalities as two inequalities each"""
    _need_setup(solver)
    Ax, bx = _lin_block(Aeq_x, beq_x, solver.nx)
    Au, bu = _lin_block(np.zeros((0, solver.nu)) if Aeq_u is None else Aeq_u, [] if beq_u is None else beq_u,
                        solver.nu)
    return set_linear_constraints(solver, np.vstack([Ax, -Ax]), np.concatenate([bx, -bx]),
                                  np.vstack([Au, -Au]), np.concatenate([bu, -bu]))


def set_cone_constraints(solver, Acu, qcu, cu, Acx, qcx, cx, *, verbose=False):
    """TinyMPC.jl:245-259.  Per-knot second-order cones, inputs first; parity unpinned."""
    ia = [np.asarray(a, dtype=np.int32) for a in (Acu, qcu, Acx, qcx)]
    da = [np.asarray(a, dtype=np.float64) for a in (cu, cx)]
    status = load_library().set_cone_constraints(
        ia[0].ctypes.data_as(c_ip), len(ia[0]), ia[1].ctypes.data_as(c_ip), len(ia[1]), _dp(da[0]),
        len(da[0]), ia[2].ctypes.data_as(c_ip), len(ia[2]), ia[3].ctypes.data_as(c_ip), len(ia[3]),
        _dp(da[1]), len(da[1]), 1 if verbose else 0)
    if status != 0:
        raise TinyMPCError(f"Failed to set cone constraints ({_err()})")
    return status


def _solve_lqr(A, B, Q, R, rho):
    """TinyMPC.jl:326-352: the rho-regularised LQR the sensitivities are differenced on (rho enters once, P0 = Q + rho I)"""
    nx, nu = A.shape[0], B.shape[1]
    Qr, Rr = Q + rho * np.eye(nx), R + rho * np.eye(nu)
    P, K = Qr.copy(), np.zeros((nu, nx))
    for it in range(1, 5001):
        Kp = K
        K = np.linalg.solve(Rr + B.T @ P @ B + 1e-8 * np.eye(nu), B.T @ P @ A)
        P = Qr + A.T @ P @ (A - B @ K)
        if it > 1 and np.linalg.norm(K - Kp) < 1e-10:
            break
    return K, P, np.linalg.inv(Rr + B.T @ P @ B), (A - B @ K).T


def compute_sensitivity_autograd(solver):
    """TinyMPC.jl:301-323: forward differences (h = 1e-6) of (Kinf, Pinf, Quu_inv, AmBKt) in rho -> (dK, dP, dC1, dC2)"""
    _need_setup(solver)
    h = 1e-6
    m0 = _solve_lqr(solver.A, solver.B, solver.Q, solver.R, solver.rho)
    m1 = _solve_lqr(solver.A, solver.B, solver.Q, solver.R, solver.rho + h)
    return tuple((b - a) / h for a, b in zip(m0, m1))


def set_sensitivity(solver, dK, dP, dC1=None, dC2=None, *, verbose=False):
    """hand the live solver the sensitivities codegen_with_sensitivity would bake in (TinyMPC.jl:374-394)"""
    _need_setup(solver)
    ms = [None if m is None else _mat(m) for m in (dK, dP, dC1, dC2)]
    args = []
    for m in ms:
        args += [None, 0, 0] if m is None else [_dp(m), m.shape[0], m.shape[1]]
    if load_library().set_sensitivity(*args, 1 if verbose else 0) != 0:
        raise TinyMPCError(f"Failed to set sensitivity matrices ({_err()})")
    return 0


def get_adaptive_rho(solver):
    """rho of each instance after adaptation, shape (batch,)"""
    _need_setup(solver)
    rho, n = np.zeros(solver.batch), ctypes.c_int()
    if load_library().get_adaptive_rho(_dp(rho), ctypes.byref(n)) != 0:
        raise TinyMPCError(f"Failed to get adaptive rho ({_err()})")
    return rho[: n.value]


def set_cache_terms(solver, Kinf, Pinf, Quu_inv, AmBKt, *, verbose=False):
    """TinyMPC.jl:278-292"""
    _need_setup(solver)
    ms = [_mat(m) for m in (Kinf, Pinf, Quu_inv, AmBKt)]
    args = []
    for m in ms:
        args += [_dp(m), m.shape[0], m.shape[1]]
    status = load_library().set_cache_terms(*args, 1 if verbose else 0)
    if status != 0:
        raise TinyMPCError(f"Failed to set cache terms ({_err()})")
    return status


def print_problem_data(solver, *, verbose=False):
    _need_setup(solver)
    return load_library().print_problem_data(1 if verbose else 0)


def cleanup():
    """TinyMPC.jl:428-429"""
    try:
        load_library().cleanup_solver()
    except Exception:
        pass


def host_sensitivity(A, B, Q, R, rho):
    """the library's host finite differences (dK, dP, dC1, dC2) without a GPU (TinyMPC.jl:301-352 semantics)"""
    A, B, Q, R = _mat(A), _mat(B), _mat(Q), _mat(R)
    nx, nu = A.shape[0], B.shape[1]
    dK, dP = np.zeros((nu, nx), order="F"), np.zeros((nx, nx), order="F")
    d1, d2 = np.zeros((nu, nu), order="F"), np.zeros((nx, nx), order="F")
    if load_library().tinympc_host_sensitivity(_dp(A), _dp(B), _dp(Q), _dp(R), float(rho), nx, nu, _dp(dK), _dp(dP),
                                               _dp(d1), _dp(d2)) != 0:
        raise TinyMPCError(f"host_sensitivity failed ({_err()})")
    return dK, dP, d1, d2


def host_precompute(A, B, Q, R, rho):
    """fp64 Riccati cache on the host (no GPU needed): dict(Kinf, Pinf, Quu_inv, AmBKt)."""
    A, B, Q, R = _mat(A), _mat(B), _mat(Q), _mat(R)
    nx, nu = A.shape[0], B.shape[1]
    K, P = np.zeros((nu, nx), order="F"), np.zeros((nx, nx), order="F")
    Qi, Am = np.zeros((nu, nu), order="F"), np.zeros((nx, nx), order="F")
    if load_library().tinympc_host_precompute(_dp(A), _dp(B), _dp(Q), _dp(R), float(rho), nx, nu, _dp(K),
                                              _dp(P), _dp(Qi), _dp(Am)) != 0:
        raise TinyMPCError(f"host_precompute failed ({_err()})")
    return dict(Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am)


class BatchSolver:
    """Handle-API wrapper (tinympc_* in include/tinympc_hip.h): several solvers per process,
    device-resident I/O, stream-ordered solves.  Used by bench.py and the sharded driver."""

    def __init__(self, A, B, Q, R, rho, N, batch, device=-1, verbose=False):
        self.lib = load_library()
        A, B, Q, R = _mat(A), _mat(B), _mat(Q), _mat(R)
        self.nx, self.nu, self.N, self.batch = A.shape[0], B.shape[1], int(N), int(batch)
        h = c_vp()
        st = self.lib.tinympc_create(ctypes.byref(h), _dp(A), _dp(B), _dp(Q), _dp(R), float(rho), self.nx,
                                     self.nu, self.N, self.batch, int(device), 1 if verbose else 0)
        if st != 0:
            raise TinyMPCError(f"tinympc_create failed ({_err()})")
        self.h = h

    @classmethod
    def from_families(cls, A, B, Q, R, rho, N, device=-1, verbose=False):
        """One (A, B, Q, R, rho) family PER INSTANCE: A (nx, nx, batch), B (nx, nu, batch), Q (nx, nx, batch),
        R (nu, nu, batch), rho (batch,).  Each instance gets its own Riccati cache (host, fp64)."""
        self = cls.__new__(cls)
        self.lib = load_library()
        A, B, Q, R = (np.asfortranarray(np.asarray(m, dtype=np.float64)) for m in (A, B, Q, R))
        rho = np.ascontiguousarray(np.asarray(rho, dtype=np.float64))
        self.nx, self.nu, self.N, self.batch = A.shape[0], B.shape[1], int(N), A.shape[2]
        h = c_vp()
        st = self.lib.tinympc_create_families(ctypes.byref(h), _dp(A), _dp(B), _dp(Q), _dp(R), _dp(rho), self.nx,
                                              self.nu, self.N, self.batch, int(device), 1 if verbose else 0)
        if st != 0:
            raise TinyMPCError(f"tinympc_create_families failed ({_err()})")
        self.h = h
        return self

    def close(self):
        if getattr(self, "h", None):
            self.lib.tinympc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, what):
        if st != 0:
            raise TinyMPCError(f"{what} failed ({_err()})")

    def update_settings(self, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1,
                        en_state_bound=0, en_input_bound=0):
        self._chk(self.lib.tinympc_update_settings(self.h, float(abs_pri_tol), float(abs_dua_tol),
                                                   int(max_iter), int(check_termination),
                                                   int(en_state_bound), int(en_input_bound)),
                  "update_settings")

    def set_bound_constraints(self, x_min, x_max, u_min, u_max):
        ms = [_mat(m) for m in (x_min, x_max, u_min, u_max)]
        self._chk(self.lib.tinympc_set_bound_constraints(self.h, *[_dp(m) for m in ms]), "set_bound_constraints")

    def set_cache_terms(self, Kinf, Pinf, Quu_inv, AmBKt):
        ms = [_mat(m) for m in (Kinf, Pinf, Quu_inv, AmBKt)]
        self._chk(self.lib.tinympc_set_cache_terms(self.h, *[_dp(m) for m in ms]), "set_cache_terms")

    def set_fdyn(self, fdyn):
        """affine dynamics term x+ = A x + B u + f (parity unpinned; stream / generic kernels)"""
        f = np.ascontiguousarray(np.asarray(fdyn, dtype=np.float64).reshape(-1))
        self._chk(self.lib.tinympc_set_fdyn(self.h, _dp(f)), "set_fdyn")

    def set_cone_constraints(self, Acu, qcu, cu, Acx, qcx, cx):
        """per-knot second-order cones, inputs first (TinyMPC.jl:245-259 argument order); parity unpinned"""
        ia = [np.ascontiguousarray(np.asarray(a, dtype=np.int32)) for a in (Acu, qcu, Acx, qcx)]
        da = [np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (cu, cx)]
        self._chk(self.lib.tinympc_set_cone_constraints(
            self.h, ia[0].ctypes.data_as(c_ip), ia[1].ctypes.data_as(c_ip), _dp(da[0]), len(da[0]),
            ia[2].ctypes.data_as(c_ip), ia[3].ctypes.data_as(c_ip), _dp(da[1]), len(da[1])), "set_cone_constraints")

    def set_linear_constraints(self, Alin_x, blin_x, Alin_u, blin_u):
        """Alin_x x <= blin_x, Alin_u u <= blin_u at every knot (TinyMPC.jl:229-243); parity unpinned"""
        Ax, bx = _lin_block(Alin_x, blin_x, self.nx)
        Au, bu = _lin_block(Alin_u, blin_u, self.nu)
        self._chk(self.lib.tinympc_set_linear_constraints(self.h, _dp(Ax), Ax.shape[0], _dp(bx), _dp(Au), Au.shape[0],
                                                          _dp(bu)), "set_linear_constraints")

    def enable_linear(self, en_state_linear, en_input_linear):
        """the two linear-row switches of update_settings (TinyMPC.jl:98-99) on their own: a side keeps its rows while it is off"""
        self._chk(self.lib.tinympc_enable_linear(self.h, int(bool(en_state_linear)), int(bool(en_input_linear))), "enable_linear")

    def set_equality_constraints(self, Aeq_x, beq_x, Aeq_u=None, beq_u=None):
        """equalities as two inequalities each (TinyMPC.jl:261-270)"""
        Ax, bx = _lin_block(Aeq_x, beq_x, self.nx)
        Au, bu = _lin_block(np.zeros((0, self.nu)) if Aeq_u is None else Aeq_u, [] if beq_u is None else beq_u, self.nu)
        self.set_linear_constraints(np.vstack([Ax, -Ax]), np.concatenate([bx, -bx]), np.vstack([Au, -Au]),
                                    np.concatenate([bu, -bu]))

    def set_adaptive_rho(self, enable=True, rho_min=0.1, rho_max=10.0, enable_clipping=True):
        """per-instance rho adaptation every 5th iteration (admm.cpp:147-174); defaults TinyMPC.jl:59-61"""
        self._chk(self.lib.tinympc_set_adaptive_rho(self.h, int(bool(enable)), float(rho_min), float(rho_max),
                                                    int(bool(enable_clipping))), "set_adaptive_rho")

    def set_sensitivity(self, dK, dP, dC1=None, dC2=None):
        """dKinf/drho, dPinf/drho (dC1, dC2 accepted and unused, as in the reference's iteration)"""
        ms = [None if m is None else _mat(m) for m in (dK, dP, dC1, dC2)]
        self._chk(self.lib.tinympc_set_sensitivity(self.h, *[None if m is None else _dp(m) for m in ms]), "set_sensitivity")

    def compute_sensitivity(self):
        """(dK, dP, dC1, dC2) by the library's host finite differences (TinyMPC.jl:301-352)"""
        nx, nu = self.nx, self.nu
        dK, dP = np.zeros((nu, nx), order="F"), np.zeros((nx, nx), order="F")
        d1, d2 = np.zeros((nu, nu), order="F"), np.zeros((nx, nx), order="F")
        self._chk(self.lib.tinympc_compute_sensitivity(self.h, _dp(dK), _dp(dP), _dp(d1), _dp(d2)), "compute_sensitivity")
        return dK, dP, d1, d2

    def get_adaptive_state(self):
        """each instance's current rho (batch,), Kinf (nu, nx, batch), Pinf (nx, nx, batch)"""
        nx, nu, Bn = self.nx, self.nu, self.batch
        rho, K, P = np.zeros(Bn), np.zeros((nu, nx, Bn), order="F"), np.zeros((nx, nx, Bn), order="F")
        self._chk(self.lib.tinympc_get_adaptive_state(self.h, _dp(rho), _dp(K), _dp(P)), "get_adaptive_state")
        return dict(rho=rho, Kinf=K, Pinf=P)

    def get_cache_terms(self):
        nx, nu = self.nx, self.nu
        K, P = np.zeros((nu, nx), order="F"), np.zeros((nx, nx), order="F")
        Qi, Am = np.zeros((nu, nu), order="F"), np.zeros((nx, nx), order="F")
        self._chk(self.lib.tinympc_get_cache_terms(self.h, _dp(K), _dp(P), _dp(Qi), _dp(Am)), "get_cache_terms")
        return dict(Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am)

    def set_x0(self, x0):
        m = _mat(x0)
        self._chk(self.lib.tinympc_set_x0(self.h, _dp(m), m.shape[1]), "set_x0")

    def set_x_ref(self, x_ref):
        m = _ref3(x_ref)
        self._chk(self.lib.tinympc_set_x_ref(self.h, _dp(m), m.shape[1]), "set_x_ref")

    def set_u_ref(self, u_ref):
        m = _ref3(u_ref)
        self._chk(self.lib.tinympc_set_u_ref(self.h, _dp(m), m.shape[1]), "set_u_ref")

    def reset(self):
        self._chk(self.lib.tinympc_reset(self.h), "reset")

    def set_warm_start(self, on):
        self._chk(self.lib.tinympc_set_warm_start(self.h, 1 if on else 0), "set_warm_start")

    def solve(self):
        st = int(self.lib.tinympc_solve(self.h))
        if st < 0:
            raise TinyMPCError(f"solve failed ({_err()})")
        return st

    def solve_async(self, stream=None):
        self._chk(self.lib.tinympc_solve_async(self.h, c_vp(stream or 0)), "solve_async")

    def solve_status(self):
        return int(self.lib.tinympc_solve_status(self.h))

    def get_solution(self):
        nx, nu, N, B = self.nx, self.nu, self.N, self.batch
        sb, cb = np.zeros(nx * N * B), np.zeros(nu * (N - 1) * B)
        self._chk(self.lib.tinympc_get_states(self.h, _dp(sb)), "get_states")
        self._chk(self.lib.tinympc_get_controls(self.h, _dp(cb)), "get_controls")
        return dict(states=sb.reshape((nx, N, B), order="F"), controls=cb.reshape((nu, N - 1, B), order="F"))

    def get_status(self):
        B = self.batch
        it, so, res = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros((B, 4))
        self._chk(self.lib.tinympc_get_status(self.h, it.ctypes.data_as(c_ip), so.ctypes.data_as(c_ip), _dp(res)),
                  "get_status")
        return dict(iter=it, solved=so, residuals=res)

    def get_workspace(self):
        nx, nu, N, B = self.nx, self.nu, self.N, self.batch
        d, y, z = (np.zeros(nu * (N - 1) * B) for _ in range(3))
        g, v = (np.zeros(nx * N * B) for _ in range(2))
        self._chk(self.lib.tinympc_get_workspace(self.h, _dp(d), _dp(y), _dp(g), _dp(v), _dp(z)), "get_workspace")
        ru = lambda a: a.reshape((nu, N - 1, B), order="F")
        rx = lambda a: a.reshape((nx, N, B), order="F")
        return dict(d=ru(d), y=ru(y), z=ru(z), g=rx(g), v=rx(v))

    def device_buffers(self):
        ptrs = [c_vp() for _ in range(9)]
        self._chk(self.lib.tinympc_device_buffers(self.h, *[ctypes.byref(p) for p in ptrs]), "device_buffers")
        names = ("x0", "x_ref", "u_ref", "states", "controls", "iter", "solved", "residuals", "gstat")
        return {n: p.value for n, p in zip(names, ptrs)}

    def set_ref_mode(self, mode):
        self._chk(self.lib.tinympc_set_ref_mode(self.h, int(mode)), "set_ref_mode")

    def set_x0_f32(self, x0):
        """x0 as a float32 array (nx,) or (nx, B): a plain copy to the device, no fp64 pass on the host"""
        m = np.asfortranarray(np.asarray(x0, dtype=np.float32))
        m = m.reshape(self.nx, -1, order="F")
        self._chk(self.lib.tinympc_set_x0_f32(self.h, m.ctypes.data_as(c_fp), m.shape[1]), "set_x0_f32")

    def get_solution_f32(self, states=None, controls=None):
        """the solution into float32 arrays (nx, N, B) / (nu, N-1, B) (allocated, or the caller's own F-ordered ones)"""
        nx, nu, N, B = self.nx, self.nu, self.N, self.batch
        states = np.zeros((nx, N, B), dtype=np.float32, order="F") if states is None else states
        controls = np.zeros((nu, N - 1, B), dtype=np.float32, order="F") if controls is None else controls
        assert states.flags.f_contiguous and controls.flags.f_contiguous and states.dtype == controls.dtype == np.float32
        self._chk(self.lib.tinympc_get_states_f32(self.h, states.ctypes.data_as(c_fp)), "get_states_f32")
        self._chk(self.lib.tinympc_get_controls_f32(self.h, controls.ctypes.data_as(c_fp)), "get_controls_f32")
        return dict(states=states, controls=controls)

    def pin_host(self, arr):
        """page-lock a numpy array the caller keeps alive and reuses (the fp32 transfers then DMA straight into it);
        call unpin_host(arr) before dropping the array"""
        self._chk(self.lib.tinympc_pin_host(self.h, arr.ctypes.data_as(c_vp), arr.nbytes), "pin_host")

    def unpin_host(self, arr):
        self._chk(self.lib.tinympc_unpin_host(self.h, arr.ctypes.data_as(c_vp)), "unpin_host")

    def set_ref_sequence(self, x_ref_seq, u_ref_seq):
        """shared references of every step of the next closed loops: x_ref_seq (nx, N, steps), u_ref_seq (nu, N-1, steps)
        (rocket_landing_constraints.jl:107-115 shifts them step by step); None, None drops the sequence"""
        if x_ref_seq is None:
            self._chk(self.lib.tinympc_set_ref_sequence(self.h, None, 0, 0, None, 0, 0, 0), "set_ref_sequence")
            return
        xs = np.asfortranarray(np.asarray(x_ref_seq, dtype=np.float64))
        us = np.asfortranarray(np.asarray(u_ref_seq, dtype=np.float64))
        steps = xs.shape[2]
        assert xs.shape == (self.nx, self.N, steps) and us.shape == (self.nu, self.N - 1, steps)
        xs, us = xs.reshape(-1, order="F"), us.reshape(-1, order="F")
        self._chk(self.lib.tinympc_set_ref_sequence(self.h, _dp(xs), self.nx, self.N * steps, _dp(us), self.nu,
                                                    (self.N - 1) * steps, steps), "set_ref_sequence")

    def mpc_rollout(self, steps, stream=None):
        """`steps` fused closed-loop MPC steps in one launch (plant = the family's own A, B).
        Returns dict(status, x=(nx, steps, B), u=(nu, steps, B), iter=(steps, B), solved=(steps, B))."""
        st = int(self.lib.tinympc_mpc_rollout(self.h, int(steps), c_vp(stream or 0)))
        if st < 0:
            raise TinyMPCError(f"mpc_rollout failed ({_err()})")
        nx, nu, B = self.nx, self.nu, self.batch
        x, u = np.zeros(nx * steps * B), np.zeros(nu * steps * B)
        it = np.zeros(steps * B, dtype=np.int32)
        self._chk(self.lib.tinympc_get_mpc_log(self.h, _dp(x), _dp(u), it.ctypes.data_as(c_ip)), "get_mpc_log")
        it = it.reshape((steps, B), order="F")
        return dict(status=st, x=x.reshape((nx, steps, B), order="F"), u=u.reshape((nu, steps, B), order="F"),
                    iter=np.abs(it), solved=(it > 0).astype(np.int32))

    def set_profiling(self, on):
        self._chk(self.lib.tinympc_set_profiling(self.h, 1 if on else 0), "set_profiling")

    def set_compaction(self, chunk_iters):
        """tolerance-terminated solves in chunks of `chunk_iters` iterations with the unconverged instances compacted
        in between (0: off)"""
        self._chk(self.lib.tinympc_set_compaction(self.h, int(chunk_iters)), "set_compaction")

    def kernel_elapsed_ms(self, last_n=1):
        """duration of the last launch, or the mean over the last `last_n` launches (profiling mode)"""
        if last_n == 1:
            return float(self.lib.tinympc_kernel_elapsed_ms(self.h))
        return float(self.lib.tinympc_kernel_elapsed_mean_ms(self.h, int(last_n)))

    def set_precision(self, precision):
        """0: fp64 recurrences, fp32 state (default); 1: all fp32; 2: all fp64 like the reference (generic kernel, slow)"""
        self._chk(self.lib.tinympc_set_precision(self.h, int(precision)), "set_precision")

    def reload_switches(self):
        """re-read the TINYMPC_HIP_* environment switches (they are read once, at creation)"""
        self._chk(self.lib.tinympc_reload_switches(self.h), "reload_switches")

    def set_strict_precision(self, strict):
        """True: precision = 1 means fp32 recurrences even where a (faster) fp64 matrix-core kernel exists for the shape"""
        self._chk(self.lib.tinympc_set_strict_precision(self.h, 1 if strict else 0), "set_strict_precision")

    @property
    def effective_precision(self):
        """recurrence precision the current options run with: 0 fp64, 1 fp32"""
        return int(self.lib.tinympc_effective_precision(self.h))

    @property
    def kernel_name(self):
        return self.lib.tinympc_kernel_name(self.h).decode()

    @property
    def last_launch_name(self):
        """the kernel the most recent launch actually ran (the family, or e.g. its lean<nx,nu,N> one-shot variant)"""
        return self.lib.tinympc_last_launch_name(self.h).decode()

    def algorithmic_bytes(self):
        return float(self.lib.tinympc_algorithmic_bytes(self.h))

    def algorithmic_flops(self, iters):
        return float(self.lib.tinympc_algorithmic_flops(self.h, int(iters)))


def specialise(nx, nu, N, verbose=False):
    """compile / load the on-chip kernel of a shape the library was not built with, ahead of time (what BatchSolver does by
    itself at creation): True if the shape has an on-chip kernel afterwards"""
    return bool(load_library().tinympc_specialise(int(nx), int(nu), int(N), 1 if verbose else 0))


def shard_range(batch, n_shards, shard):
    """the library's partition rule (no GPU needed): [lo, hi) of `shard`"""
    lo, hi = c_int(), c_int()
    load_library().tinympc_shard_range(int(batch), int(n_shards), int(shard), ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


class ShardedBatchSolver:
    """One handle, several GPUs of one node, ONE host process (tinympc_sharded_* in include/tinympc_hip.h): the batch in
    contiguous shards, one per device; inputs scattered, outputs gathered, the solve status all-reduced over RCCL."""

    def __init__(self, A, B, Q, R, rho, N, batch, n_gpus=1, devices=None, verbose=False):
        self.lib = load_library()
        A, B, Q, R = _mat(A), _mat(B), _mat(Q), _mat(R)
        self.nx, self.nu, self.N, self.batch = A.shape[0], B.shape[1], int(N), int(batch)
        dv = None
        if devices is not None:
            devices = np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
            n_gpus, dv = len(devices), devices.ctypes.data_as(c_ip)
        h = c_vp()
        st = self.lib.tinympc_create_sharded(ctypes.byref(h), _dp(A), _dp(B), _dp(Q), _dp(R), float(rho), self.nx, self.nu,
                                             self.N, self.batch, int(n_gpus), dv, 1 if verbose else 0)
        if st != 0:
            raise TinyMPCError(f"tinympc_create_sharded failed ({_err()})")
        self.h = h
        self.n_shards = int(self.lib.tinympc_sharded_n_shards(h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.tinympc_sharded_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, what):
        if st != 0:
            raise TinyMPCError(f"{what} failed ({_err()})")

    @property
    def fold_backend(self):
        return self.lib.tinympc_sharded_fold_backend(self.h).decode()

    def shard(self, i):
        """(device, lo, hi, handle of the shard's single-device solver)"""
        d, lo, hi, loc = c_int(), c_int(), c_int(), c_vp()
        self._chk(self.lib.tinympc_sharded_shard(self.h, int(i), ctypes.byref(d), ctypes.byref(lo), ctypes.byref(hi),
                                                 ctypes.byref(loc)), "shard")
        return d.value, lo.value, hi.value, loc

    def kernel_names(self):
        return [self.lib.tinympc_kernel_name(self.shard(i)[3]).decode() for i in range(self.n_shards)]

    def update_settings(self, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=0,
                        en_input_bound=0):
        self._chk(self.lib.tinympc_sharded_update_settings(self.h, float(abs_pri_tol), float(abs_dua_tol), int(max_iter),
                                                           int(check_termination), int(en_state_bound),
                                                           int(en_input_bound)), "update_settings")

    def set_bound_constraints(self, x_min, x_max, u_min, u_max):
        ms = [_mat(m) for m in (x_min, x_max, u_min, u_max)]
        self._chk(self.lib.tinympc_sharded_set_bound_constraints(self.h, *[_dp(m) for m in ms]), "set_bound_constraints")

    def set_warm_start(self, on):
        self._chk(self.lib.tinympc_sharded_set_warm_start(self.h, 1 if on else 0), "set_warm_start")

    def reset(self):
        self._chk(self.lib.tinympc_sharded_reset(self.h), "reset")

    def set_precision(self, precision):
        self._chk(self.lib.tinympc_sharded_set_precision(self.h, int(precision)), "set_precision")

    def set_compaction(self, chunk_iters):
        self._chk(self.lib.tinympc_sharded_set_compaction(self.h, int(chunk_iters)), "set_compaction")

    def set_x0(self, x0):
        m = _mat(x0)
        self._chk(self.lib.tinympc_sharded_set_x0(self.h, _dp(m), m.shape[1]), "set_x0")

    def set_x_ref(self, x_ref):
        m = _ref3(x_ref)
        self._chk(self.lib.tinympc_sharded_set_x_ref(self.h, _dp(m), m.shape[1]), "set_x_ref")

    def set_u_ref(self, u_ref):
        m = _ref3(u_ref)
        self._chk(self.lib.tinympc_sharded_set_u_ref(self.h, _dp(m), m.shape[1]), "set_u_ref")

    def solve(self):
        st = int(self.lib.tinympc_sharded_solve(self.h))
        if st < 0:
            raise TinyMPCError(f"sharded solve failed ({_err()})")
        return st

    def solve_async(self):
        self._chk(self.lib.tinympc_sharded_solve_async(self.h), "solve_async")

    def wait(self):
        st = int(self.lib.tinympc_sharded_wait(self.h))
        if st < 0:
            raise TinyMPCError(f"sharded wait failed ({_err()})")
        return st

    def global_status(self):
        """(residual maxima over all instances (4,), largest per-device unsolved count) of the last solve"""
        res, n = np.zeros(4), c_int()
        self._chk(self.lib.tinympc_sharded_global_status(self.h, _dp(res), ctypes.byref(n)), "global_status")
        return res, n.value

    def get_solution(self):
        nx, nu, N, B = self.nx, self.nu, self.N, self.batch
        sb, cb = np.zeros(nx * N * B), np.zeros(nu * (N - 1) * B)
        self._chk(self.lib.tinympc_sharded_get_states(self.h, _dp(sb)), "get_states")
        self._chk(self.lib.tinympc_sharded_get_controls(self.h, _dp(cb)), "get_controls")
        return dict(states=sb.reshape((nx, N, B), order="F"), controls=cb.reshape((nu, N - 1, B), order="F"))

    def get_status(self):
        B = self.batch
        it, so, res = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros((B, 4))
        self._chk(self.lib.tinympc_sharded_get_status(self.h, it.ctypes.data_as(c_ip), so.ctypes.data_as(c_ip), _dp(res)),
                  "get_status")
        return dict(iter=it, solved=so, residuals=res)

    def get_workspace(self):
        nx, nu, N, B = self.nx, self.nu, self.N, self.batch
        d, y, z = (np.zeros(nu * (N - 1) * B) for _ in range(3))
        g, v = (np.zeros(nx * N * B) for _ in range(2))
        self._chk(self.lib.tinympc_sharded_get_workspace(self.h, _dp(d), _dp(y), _dp(g), _dp(v), _dp(z)), "get_workspace")
        ru = lambda a: a.reshape((nu, N - 1, B), order="F")
        rx = lambda a: a.reshape((nx, N, B), order="F")
        return dict(d=ru(d), y=ru(y), z=ru(z), g=rx(g), v=rx(v))
