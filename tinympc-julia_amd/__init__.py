"""tinympc-julia_amd — MI355X-native batched TinyMPC ADMM engine.

Contents (only what the hot path needs):
  csrc/        hand-written HIP kernels for gfx950 + the C-ABI shim (libtinympc_hip.so)
  lib/         the built shared library (git-ignored, travels with gpurun snapshots)
  julia/       TinyMPC.jl — the reference's Julia surface with a batch dimension (ccall host)
  tinympc.py   the same surface in Python over ctypes (the Julia toolchain is absent from
               this image; tests drive the C-ABI through this mirror)
  sharding.py  one-process-per-GPU batch sharding + the RCCL status all-reduce
  problems.py  problem families / synthetic batches of the benchmark configs

The compute path is the HIP library only: there is no CPU fallback, and nothing here
imports oracle/.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtinympc_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tinympc_hip.h")


def build(jobs=8, verbose=False):
    """Compile csrc/ for gfx950 with hipcc into lib/libtinympc_hip.so (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


from . import problems  # noqa: E402,F401
from .tinympc import (  # noqa: E402,F401
    BatchSolver,
    ShardedBatchSolver,
    set_gpus,
    shard_range,
    TinyMPCError,
    TinyMPCSolver,
    cleanup,
    compute_sensitivity_autograd,
    get_solution,
    get_status,
    host_precompute,
    host_sensitivity,
    load_library,
    print_problem_data,
    reset_workspace,
    set_batch_size,
    set_bound_constraints,
    set_cache_terms,
    set_cone_constraints,
    set_equality_constraints,
    set_linear_constraints,
    set_sensitivity,
    get_adaptive_rho,
    set_u_ref,
    set_x0,
    set_x_ref,
    setup,
    solve,
    update_settings,
)
