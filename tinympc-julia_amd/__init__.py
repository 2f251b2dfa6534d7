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


STAMP_PATH = os.path.join(_HERE, "lib", "build_stamp.json")


def source_hash():
    """sha256 over the library's sources (csrc/*, include/tinympc_hip.h): what the built .so is stamped with"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + [os.path.join(_HERE, "csrc", "Makefile"), HEADER_PATH])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build(jobs=8, verbose=False):
    """Compile csrc/ for gfx950 with hipcc into lib/libtinympc_hip.so (cross-compiles without a GPU).  `make` is
    incremental (seconds when nothing changed); the result is stamped with the hash of the sources it was built from,
    the wall time of the make and the compiler version (lib/build_stamp.json)."""
    import json
    import time
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"]
    if not verbose:
        cmd.insert(1, "-s")
    before = os.path.getmtime(LIB_PATH) if os.path.isfile(LIB_PATH) else None
    t0 = time.time()
    subprocess.check_call(cmd)
    secs = time.time() - t0
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f"build did not produce {LIB_PATH}")
    relinked = before is None or os.path.getmtime(LIB_PATH) != before
    try:
        stale = json.load(open(STAMP_PATH))["source_sha256"] != source_hash()
    except Exception:
        stale = True
    if relinked or stale:
        try:
            ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()
        except Exception:
            ver = []
        json.dump({"source_sha256": source_hash(), "make_seconds": round(secs, 1), "jobs": jobs,
                   "from_scratch": before is None, "hipcc_version": [l for l in ver if l.strip()][:3],
                   "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                   "host_cpus": os.cpu_count()}, open(STAMP_PATH, "w"), indent=1)
    return LIB_PATH


def ensure_built(jobs=8):
    """The library, rebuilt from source whenever it is missing or was built from other sources than the ones present
    (hash in lib/build_stamp.json — file times do not survive a repository snapshot)."""
    import json
    try:
        fresh = os.path.isfile(LIB_PATH) and json.load(open(STAMP_PATH))["source_sha256"] == source_hash()
    except Exception:
        fresh = False
    return LIB_PATH if fresh else build(jobs)


from . import problems  # noqa: E402,F401
from .tinympc import (  # noqa: E402,F401
    BatchSolver,
    ShardedBatchSolver,
    set_gpus,
    set_warm_start,
    kernel_name,
    shard_range,
    specialise,
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
    set_ref_sequence,
    mpc_rollout,
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
