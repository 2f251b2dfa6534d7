#!/usr/bin/env python3
"""Benchmark of the batched TinyMPC ADMM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one batched solve of the workload on every rank.  Default workload = BASELINE.json configs[1]:
cartpole nx=4 nu=1 N=20, u in [-0.5, 0.5], batch 65 536 per GPU, cold start, exactly 100 ADMM iterations per
instance (tolerances 0); scaling = weak (per-GPU batch fixed).  `--scaling strong` = configs[4]: quadrotor
nx=12 nu=4 N=30, 2^20 instances IN TOTAL (seed 3), rank r solving the contiguous shard
sharding.shard_range(2^20, r, N).  Inputs are resident in HBM before the timed region.

One process per GPU.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher:
before anything touches a GPU it starts `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`
as a child process and exits with its code (the ranks are fresh processes; nothing is re-exec'd).  Under a launcher
(WORLD_SIZE set) the process is a rank and asserts WORLD_SIZE == --gpus.

The batch shards with no data-path collective.  The path's only exchange is an all-reduce(MAX) over RCCL of the
5-word status block (4 residual maxima + unsolved count) that decides the global solve status: the fixed-iteration
workload needs it once, after the last solve; `--status-every-step` (and every tolerance-terminated
run, `--tol`) does it after every solve, overlapped with the next one.  Rank 0 prints ONE JSON line.

Before the W warm-up steps the GPU clocks are ramped with 150 ms of untimed solves (set-up, like building the
solver); the timed region is exactly K steps, bracketed by barrier + synchronize, max over ranks.

On one GPU the same line also carries `configs`: five timed steps each of BASELINE configs[2] (quadrotor 65 536),
configs[3] (rocket + cones 32 768) and one rank's shard of configs[4] (quadrotor 131 072, tolerance-terminated),
each with its kernel time and roofline (`--no-extras` skips them), and `cpu_baseline`.

`--dry --backend gloo` rehearses the launch / sharding / status-fold path on CPUs (no GPU, no solver): the CPU test
of the N > 1 launch.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _lib_hash():
    """source hash of the library (csrc/ + include/), the same digest as tinympc_julia_amd.source_hash() — restated here
    because the parent of a multi-rank launch must not import the package (tests/test_bench_launch.py holds them equal)"""
    import glob
    import hashlib
    pkg = os.path.join(ROOT, "tinympc-julia_amd")
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")) +
                   glob.glob(os.path.join(pkg, "csrc", "*.cpp")) +
                   [os.path.join(pkg, "csrc", "Makefile"), os.path.join(ROOT, "include", "tinympc_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


LIB_HASH = _lib_hash()
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector
FP64_PEAK_TFLOPS = 78.6       # AMD MI355X spec: fp64 vector = dense fp64 matrix rate (half the FP32 rate of the guide's
                              # table, 32 FLOP/clk/SIMD; the guide has no fp64 row of its own)
STRONG_TOTAL = 1 << 20        # BASELINE configs[4]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the config's batch on every GPU (default).  strong: BASELINE configs[4], quadrotor, 2^20 "
                         "instances in total, contiguous shards")
    ap.add_argument("--config", default=None, choices=["cartpole", "quadrotor", "rocket", "rocket_soc"])
    ap.add_argument("--batch", type=int, default=0, help="weak: per-GPU batch; strong: TOTAL batch (default: the config's)")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--tol", type=float, default=0.0,
                    help="> 0: tolerance-terminated solves (abs_pri_tol = abs_dua_tol = tol) instead of fixed iterations")
    ap.add_argument("--check-termination", type=int, default=0, help="check interval (default 1; 10 with --tol)")
    ap.add_argument("--compaction", type=int, default=0, help="tolerance-terminated: chunk size of tinympc_set_compaction")
    ap.add_argument("--precision", type=int, default=0, help="0: fp64 recurrences (default), 1: all fp32")
    ap.add_argument("--keep-workspace", action="store_true",
                    help="the reference's default calling pattern: the workspace persists between solves (admm.cpp:111-115) instead of cold one-shot solves")
    ap.add_argument("--mode", default="solve", choices=["solve", "mpc"],
                    help="solve: the headline.  mpc: additionally the warm-started closed loop (SURVEY 8f)")
    ap.add_argument("--mpc-steps", type=int, default=50)
    ap.add_argument("--mpc-max-iter", type=int, default=10)
    ap.add_argument("--status-every-step", action="store_true",
                    help="N > 1: all-reduce the status block after every solve (always on with --tol)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the `configs` block (configs 3, 4, 5-shard)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sharded-capi", action="store_true",
                    help="ONE process drives all --gpus devices through the in-process multi-GPU handle of the C-ABI "
                         "(tinympc_create_sharded: what a Julia host uses) instead of one process per GPU")
    ap.add_argument("--devices", default=None,
                    help="--sharded-capi: comma-separated device of every shard (default 0..gpus-1); repeating a device, "
                         "e.g. 0,0, rehearses the multi-shard path on a one-GPU box (host status fold)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo only with --dry")
    ap.add_argument("--dry", action="store_true", help="no GPU, no solver: rehearse launch, sharding and status fold")
    ap.add_argument("--specialised-only", action="store_true",
                    help="print the `specialised` block alone (what the default run starts as a child process)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`--gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` and relay its exit code.
    Nothing in this process has touched (or will touch) a GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------------------------------
def default_batch(name):
    return 32768 if name.startswith("rocket") else 65536


def make_workload(t, name, batch, seed, lo=0, hi=None):
    """problem family, x0 (nx, hi-lo) = columns [lo, hi) of the seeded batch of `batch` instances, shared references"""
    import numpy as np
    P = t.problems
    hi = batch if hi is None else hi
    if name == "cartpole":
        prob, x0, refs = P.cartpole(20, u_bound=0.5), P.cartpole_x0(batch, seed=seed), None
        label = "cartpole nx=4 nu=1 N=20 box-only"
    elif name == "quadrotor":
        prob, x0, refs = P.quadrotor(30, u_bound=0.5), P.quadrotor_x0(batch, seed=seed), None
        label = "quadrotor nx=12 nu=4 N=30 box"
    else:
        prob, x0, refs = P.rocket(50), P.rocket_x0(batch, seed=seed), P.rocket_refs(50)
        label = "rocket nx=6 nu=3 N=50 box-only sub-problem (no fdyn/SOC)"
        if name == "rocket_soc":
            label = "rocket nx=6 nu=3 N=50 SOC thrust/glide cones + box + fdyn (parity pinned to the oracle only)"
    return prob, np.asfortranarray(x0[:, lo:hi]), refs, label


def pattern_of(tol, keep_workspace=False, adaptive=False, state_bound=None):
    """the calling pattern a counter pass belongs to (profiles/traffic.json `pattern`)"""
    if adaptive:
        return "adaptive_rho"
    if state_bound is not None:
        return "state_bound"
    if keep_workspace:
        return "workspace_kept"
    return "cold" if tol <= 0 else ("check_live" if tol < 1e-20 else "tol")


def build_solver(t, name, prob, x0, refs, device, iters, tol, check, precision, compaction=0, keep_workspace=False):
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=x0.shape[1], device=device)
    bs.update_settings(abs_pri_tol=tol, abs_dua_tol=tol, max_iter=iters, check_termination=check)
    bs._bench_check, bs._bench_tol = check, tol
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_precision(precision)
    if name == "rocket_soc":
        bs.set_fdyn(prob.fdyn)
        bs.set_cone_constraints([0], [3], [prob.extra["cone_mu_u"]], [0], [3], [prob.extra["cone_mu_x"]])
    bs.set_warm_start(bool(keep_workspace))   # default: cold start, no state I/O: compulsory traffic only
    if compaction > 0:
        bs.set_compaction(compaction)
    bs.set_x0(x0)                     # H2D once; inputs stay resident in HBM
    if refs is not None:
        bs.set_x_ref(refs[0])
        bs.set_u_ref(refs[1])
    bs.set_profiling(True)
    return bs


# ----------------------------------------------------------------------------------------------------------------------
# committed counter measurements (static: NOT measured in this run)
# ----------------------------------------------------------------------------------------------------------------------
def committed_counters(family, precision, batch, kernel, pattern="cold"):
    """rocprofv3 --pmc results committed under profiles/ for exactly this (family, batch, kernel): HBM bytes per launch
    (separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled as the gfx950 guide prescribes) and the SQ issue counters.
    They are replayed from files, not measured here; each carries its source so a stale number can be told."""
    import glob
    out = {"traffic": None, "valu_issue": None, "mfma_issue": None, "valu_insts": None, "source": None}
    path = os.path.join(ROOT, "profiles", "traffic.json")

    def lib_note(e):
        """was the profile taken on the library as it is now?  (source hash of csrc/ + include/, recorded by collect_profiles)"""
        h = e.get("library_sha256")
        if not h:
            return "library version not recorded"
        return "same library sources" if h == LIB_HASH else "taken on an EARLIER version of the library sources"

    if os.path.isfile(path):
        for e in json.load(open(path)):
            if (e["family"], e["precision"], e["batch"], e["kernel"], e.get("pattern", "cold")) == (family, precision, batch, kernel, pattern):
                out["traffic"] = e["hbm_bytes_per_launch"]
                out["source"] = ("profiles/traffic.json (committed rocprofv3 --pmc passes, " + e.get("tag", "r01") + "; " +
                                 lib_note(e) + ")")
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{family}*_sq_counters.json"))):
        e = json.load(open(p))
        if (e.get("kernel"), e.get("batch"), e.get("pattern", "cold")) == (kernel, batch, pattern) and precision == 0:
            out["valu_issue"] = e.get("derived_valu_issue_utilisation")
            out["mfma_issue"] = e.get("derived_mfma_issue_utilisation")
            out["valu_insts"] = e.get("SQ_INSTS_VALU")
            out["sq_source"] = "profiles/" + os.path.basename(p) + " (committed; " + lib_note(e) + ")"
    return out


def bs_check_interval(bs):
    return int(getattr(bs, "_bench_check", 1))


def bs_can_converge(bs):
    return bool(getattr(bs, "_bench_tol", 0.0) > 0.0)


def roofline_of(bs, family, precision, batch, iters_done, k_ms, pattern="cold"):
    """roofline + valu objects of one configuration from its kernel time.  Bytes and FLOPs are SURVEY 8(d)'s
    algorithmic figures (state on chip) x the instances x iterations of one launch."""
    alg_bytes = bs.algorithmic_bytes()
    alg_flops = bs.algorithmic_flops(1) * iters_done
    sec = k_ms * 1e-3
    ach_gbs, ach_tf = alg_bytes / sec / 1e9, alg_flops / sec / 1e12
    launched = bs.last_launch_name            # the family's kernel or the variant this calling pattern took (e.g. lean<4,1,20>)
    cc = committed_counters(family, precision, batch, launched, pattern)
    hbm = {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS}
    # What bounds the path is instruction issue (~430-950 FLOP/B against a machine balance of ~20), so the
    # dominant kernel is priced in FLOP/s.  `frac` keeps SURVEY 8(d)'s definition for every round — algorithmic FLOPs (all of
    # them, fp32 elementwise included) over the 157.3 TFLOP/s fp32 vector peak — so rounds stay comparable whatever unit
    # executes the recurrences; the mandated HBM figure (algorithmic bytes over the kernel time) is `hbm`.
    nx, nu, N = bs.nx, bs.nu, bs.N
    matvec_flops = (2.0 * (N - 1) * (2 * nx * nx + 4 * nx * nu + nu * nu) + 2.0 * nx * nx) * batch * iters_done
    roof = {"bound": "valu_fp64" if precision == 0 else "valu_fp32", "achieved": ach_tf, "peak": FP32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": ach_tf / FP32_PEAK_TFLOPS, "hbm": hbm,
            "traffic": cc["traffic"], "traffic_source": cc["source"], "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": "compute-bound path: algorithmic FLOPs over the fp32 vector peak (SURVEY 8d's definition, all rounds); "
                    "`hbm` = algorithmic bytes over the kernel time against 8 TB/s"}
    valu = {"achieved_tflops": ach_tf, "peak_tflops": FP32_PEAK_TFLOPS, "frac": ach_tf / FP32_PEAK_TFLOPS,
            "frac_of_fp64_vector_peak": ach_tf / FP64_PEAK_TFLOPS,
            # the fp64 FMAs the two recurrences cannot do without (mat-vec FLOPs of SURVEY 8a rows a2 + a7 + the terminal
            # product) over the 78.6 TFLOP/s fp64 rate: what the VALU spends on necessary work
            "necessary_fma_frac": matvec_flops / sec / 1e12 / FP64_PEAK_TFLOPS,
            "necessary_fma_flops_per_launch": matvec_flops,
            "algorithmic_flops_per_launch": alg_flops,
            "issue_utilisation": cc["valu_issue"], "counters_source": cc.get("sq_source")}
    # FLOPs of the phases the kernel that ran actually executes (SURVEY 8a's rows, the same counting as algorithmic_flops):
    # a variant may elide work that cannot change the result — the lean kernel has no state clamp / state dual (no active
    # state bound: 3 + 2 per state element less), forms the linear cost without the products by rho, and both kernels skip the
    # residual maxima (6 per element) on iterations whose check cannot matter
    E = nx * N + nu * (N - 1)
    per_iter = {"a2_forward_pass": 2.0 * (N - 1) * (nx * nx + 2 * nx * nu),
                "a7_backward_pass_grad": 2.0 * (N - 1) * (nx * nx + 2 * nx * nu + nu * nu) + (N - 1) * (nu + 2 * nx),
                "a3_update_slack": 3.0 * E, "a4_update_dual": 2.0 * E, "a5_update_linear_cost": 4.0 * E + 2.0 * nx * nx + 3 * nx,
                "a6_termination_condition": 6.0 * E}
    executed = dict(per_iter)
    check_every = bs_check_interval(bs)
    can_converge = bs_can_converge(bs)
    res_share = (1.0 / check_every if can_converge else 1.0 / max(1.0, iters_done)) if check_every > 0 else 0.0
    executed["a6_termination_condition"] *= res_share
    if launched.startswith("lean"):
        executed["a3_update_slack"] = 3.0 * nu * (N - 1)                 # input clamp only: vnew = x
        executed["a4_update_dual"] = 2.0 * nu * (N - 1)                  # y only: g stays 0
        executed["a5_update_linear_cost"] = 1.0 * nu * (N - 1)           # r~ = znew - y; q~ = x itself
        executed["a6_termination_condition"] = (2.0 * nx * (N - 1) + 4.0 * nu * (N - 1)) * res_share
        executed["a7_backward_pass_grad"] = 2.0 * (N - 1) * (nx * nx + 2 * nx * nu + nu * nu) - 2.0 * nx * (nx + nu)   # p_0 is never formed
    ex_flops = sum(executed.values()) * batch * iters_done
    roof["executed_frac"] = ex_flops / sec / 1e12 / roof["peak"]
    roof["executed_flops_per_launch"] = ex_flops
    roof["executed_note"] = ("FLOPs of the phases this kernel variant executes (elided: " +
                             ("state clamp / state dual / products by rho (no active state bound), " if launched.startswith("lean") else "") +
                             f"residual maxima on {100.0 * (1.0 - res_share):.0f} % of the iterations) over the same peak as `frac`")
    if cc["valu_insts"]:
        # upper bound of what the VALU executed: every wave64 VALU instruction counted as one FMA on 64 lanes
        valu["executed_flops_upper_bound"] = cc["valu_insts"] * 128.0
        valu["executed_frac_of_fp64_vector_peak"] = cc["valu_insts"] * 128.0 / sec / 1e12 / FP64_PEAK_TFLOPS
    if launched.startswith("mfma"):
        roof.update({"bound": "mfma", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach_tf / FP64_PEAK_TFLOPS, "frac_of_fp32_vector_peak": ach_tf / FP32_PEAK_TFLOPS,
                     "issue_utilisation": cc["mfma_issue"],
                     "counters_source": cc.get("sq_source"),
                     "note": "algorithmic FLOPs (SURVEY 8d) over the dense fp64 matrix-core peak; tiles are padded, so the "
                             "issued MFMA FLOPs are higher (DESIGN.md, mfma kernel)"})
    if launched.startswith("stream"):
        roof.update({"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS})
        roof["note"] = ("run-time-horizon kernel: the per-instance trajectories stream through HBM once per ADMM iteration "
                        "(traffic >> algorithmic bytes, by design of that kernel; DESIGN.md 3.2)")
    roof["executed_frac"] = roof["executed_flops_per_launch"] / sec / 1e12 / (roof["peak"] if roof["unit"] == "TFLOP/s" else FP32_PEAK_TFLOPS)
    roof["kernel"] = launched
    return roof, valu


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------------------------------------
def effective_cpus():
    """CPUs this process can really use: min(affinity mask, cgroup CPU quota).  A GPU box shows 256 logical CPUs in the
    affinity mask while the container's cgroup grants a 16-CPU share; threads beyond the quota only time-slice."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    eff = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {"os_cpu_count": os.cpu_count(), "sched_getaffinity": aff, "cgroup_cpu_quota": quota, "effective": eff}


def cpu_baseline(prob, x0, refs, iters, seconds):
    """The reference's own compiled snapshot (oracle/_ref) when its prebuilt .so is present, else our C restatement,
    on a bounded sample of the same workload: first one thread (the per-thread rate), then one solver per effective
    CPU.  `value` is the multi-thread rate."""
    from oracle import cpu_oracle
    kind = "ref" if cpu_oracle.have_ref() else "orc64"
    if kind == "orc64" and not os.path.isfile(cpu_oracle.PORT_LIB):
        cpu_oracle.build(port=True, ref=False)
    cpus = effective_cpus()
    nthreads = cpus["effective"]
    xr, ur = refs if refs is not None else (None, None)

    def run(nt, n, budget):
        done, spent = 0, 0.0
        while spent < budget and done < 64 * x0.shape[1]:
            r = cpu_oracle.solve_batch(kind, prob, x0[:, :n], xref=xr, uref=ur, abs_pri_tol=0.0, abs_dua_tol=0.0,
                                       max_iter=iters, nthreads=nt, want_outputs=False)
            done += n
            spent += r["seconds"]
        return done, spent

    d1, s1 = run(1, min(x0.shape[1], 2048), min(3.0, seconds / 4))
    per_thread = d1 / s1
    # each call starts its threads afresh: size a call at >= ~1 s of work so that start-up stays < 1 %
    n = int(min(x0.shape[1], max(2048, per_thread * nthreads)))
    dn, sn = run(nthreads, n, seconds)
    value = dn / sn
    what = "compiled reference snapshot, oracle/_ref" if kind == "ref" else "fp64 C restatement, oracle/"
    return {"value": value, "unit": "solves/s", "cores": nthreads, "kind": "reference" if kind == "ref" else "port",
            "per_thread": per_thread, "threads_effective": nthreads, "parallel_efficiency": value / (per_thread * nthreads),
            "cpus": cpus,
            "sample": f"{dn} cold-start solves of the same workload ({iters} fixed iters) in {sn:.1f} s on {nthreads} threads "
                      f"({n} per call), after {d1} solves in {s1:.1f} s on 1 thread ({what})"}


# ----------------------------------------------------------------------------------------------------------------------
# extra configurations (one GPU): configs 3, 4 and one shard of config 5
# ----------------------------------------------------------------------------------------------------------------------
def time_config(t, torch, dev, stream, name, batch, seed, iters=100, tol=0.0, check=1, steps=5, warmup=2, compaction=0,
                adaptive=False, keep_workspace=False, state_bound=None):
    import numpy as np
    prob, x0, refs, label = make_workload(t, name, batch, seed)
    if state_bound is not None:       # a finite bound on state row 0: the state clamp and the state dual are live (a3 / a4 at full cost)
        prob.x_max = prob.x_max.copy()
        prob.x_min = prob.x_min.copy()
        prob.x_max[0, :], prob.x_min[0, :] = state_bound, -state_bound
        label += f", |x_0| <= {state_bound:g} at every knot"
    bs = build_solver(t, name, prob, x0, refs, dev.index, iters, tol, check, 0, compaction)
    if adaptive:
        bs.set_adaptive_rho(True)
        label += ", adaptive rho (every 5th iteration)"
    if keep_workspace:
        bs.set_warm_start(True)       # the reference's default: solve() resets counters only (admm.cpp:111-115)
    try:
        def one():    # (adaptive: the adapted rho / Kinf / Pinf persist from solve to solve, as in the reference)
            bs.solve_async(stream.cuda_stream)
        # clock pre-warm, as for the headline (set-up, not a step): the first milliseconds after idle run slow while the clocks ramp
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.15:
            one()
            torch.cuda.synchronize(dev)
        for _ in range(warmup):
            one()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t0) / steps
        st = bs.get_status()
        it_mean = float(np.mean(st["iter"]))
        if tol <= 0.0:
            assert int(st["iter"].min()) == iters == int(st["iter"].max()), "work skipped"
        launches_per_step = 1
        k_ms = bs.kernel_elapsed_ms(steps)
        if compaction > 0 and tol > 0.0:
            k_ms, launches_per_step = ms, None      # several launches + compaction kernels per solve: wall time is the figure
        pat = pattern_of(tol, keep_workspace, adaptive, state_bound)
        roof, valu = roofline_of(bs, name, 0, batch, it_mean, k_ms if k_ms > 0 else ms, pat)
        roof["pattern"] = pat
        if roof["traffic"] is None:
            roof["traffic_source"] = "not profiled in this calling pattern"
        out = {"workload": f"{label}, batch={batch}, " + (f"tol={tol:g} check every {check}, max_iter={iters}" if tol > 0
                                                         else f"fixed {iters} ADMM iters") +
                           (", workspace kept between solves (warm start)" if keep_workspace else ", cold start"),
               "ms_per_step": ms, "solves_per_sec": batch / (ms * 1e-3), "kernel": bs.last_launch_name, "kernel_family": bs.kernel_name, "kernel_ms": k_ms,
               "mean_iters": it_mean, "unsolved": int((st["solved"] == 0).sum()) if tol > 0 else None,
               "roofline": roof, "valu": valu, "steps": steps}
        if launches_per_step is None:
            out["note"] = f"chunked solve with compaction every {compaction} iterations: kernel_ms = wall time per solve"
        return out
    finally:
        bs.close()


def time_specialised(t, torch, dev, stream, steps=5, warmup=2):
    """What the library compiles on request (csrc/jit.cpp) for solvers its built-in kernels do not fit, each timed beside the
    same solver with the specialisation off (TINYMPC_HIP_NO_JIT=1, read when a solver is created): a shape without a built-in
    instantiation, a constraint layout (cone lists / linear rows, bindings.cpp:414-490) the built-in entries do not compile,
    precision 2 on the headline shape.  Compile times are one-off (cached on disk) and outside the timed region."""
    import numpy as np
    P = t.problems
    os.environ.setdefault("TINYMPC_HIP_CACHE", os.path.join(ROOT, "gpurun_out", "jit_cache"))

    def rocket20():
        prob = P.rocket(20)
        def cfg(bs):
            bs.set_fdyn(prob.fdyn)
            bs.set_cone_constraints([0], [3], [0.25], [0, 3], [3, 3], [0.5, 1.5])
            bs.set_linear_constraints(np.array([[0.0, 0.0, -1.0, 0.0, 0.0, 0.3]]), [0.5], np.zeros((0, 3)), [])
        return prob, P.rocket_x0(32768, seed=2), P.rocket_refs(20), cfg, 0

    cases = {
        "cartpole_N12_65536": lambda: (P.cartpole(12, u_bound=0.5), P.cartpole_x0(65536, seed=0), None, None, 0),
        "quadrotor_N12_65536": lambda: (P.quadrotor(12, u_bound=0.5), P.quadrotor_x0(65536, seed=1), None, None, 0),
        "rocket_N20_two_state_cones_thrust_cone_one_row_32768": rocket20,
        "cartpole_65536_precision2": lambda: (P.cartpole(20, u_bound=0.5), P.cartpole_x0(65536, seed=0), None, None, 2),
    }
    out = {}
    for key, make in cases.items():
        entry = {}
        for mode in ("specialised", "without"):
            if mode == "without":
                os.environ["TINYMPC_HIP_NO_JIT"] = "1"
            try:
                prob, x0, refs, cfg, precision = make()
                t0 = time.perf_counter()
                bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=x0.shape[1], device=dev.index)
                bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
                bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
                if cfg is not None:
                    cfg(bs)
                bs.set_precision(precision)
                bs.set_warm_start(False)
                bs.set_x0(x0)
                if refs is not None:
                    bs.set_x_ref(refs[0])
                    bs.set_u_ref(refs[1])
                bs.set_profiling(True)
                bs.solve_async(stream.cuda_stream)          # (first solve: where a layout / variant is compiled)
                torch.cuda.synchronize(dev)
                setup_s = time.perf_counter() - t0
                for _ in range(warmup):
                    bs.solve_async(stream.cuda_stream)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(steps):
                    bs.solve_async(stream.cuda_stream)
                torch.cuda.synchronize(dev)
                ms = 1e3 * (time.perf_counter() - t1) / steps
                st = bs.get_status()
                assert int(st["iter"].min()) == 100 == int(st["iter"].max()), "work skipped"
                rec = {"kernel": bs.last_launch_name, "kernel_family": bs.kernel_name, "ms_per_step": ms, "kernel_ms": bs.kernel_elapsed_ms(steps),
                       "solves_per_sec": x0.shape[1] / (ms * 1e-3), "setup_and_first_solve_s": round(setup_s, 2)}
                bs.close()
            finally:
                os.environ.pop("TINYMPC_HIP_NO_JIT", None)
            if mode == "specialised":
                entry.update(rec)
                entry["workload"] = f"{key}: 100 fixed ADMM iterations, cold one-shot" + (", precision 2 (fp64 state end to end)" if precision == 2 else "")
            else:
                entry["without_specialisation"] = rec
        out[key] = entry
    return out


def run_mpc_mode(args, bs, prob, x0, dev, torch):
    """Extra (not the headline): the closed-loop regime of SURVEY 8(f) — warm-started solves, max_iter 10,
    tol 1e-3 — as K launches of one step (workspace round-trips through HBM every step) and as one fused
    launch of K steps (workspace stays on chip)."""
    K = args.mpc_steps
    bs.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=args.mpc_max_iter, check_termination=1,
                       en_state_bound=1, en_input_bound=1)
    bs.set_warm_start(True)
    res = {}
    for label, launches, steps in (("one_step_per_launch", K, 1), ("fused", 1, K)):
        bs.reset()
        bs.set_x0(x0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(launches):
            bs.lib.tinympc_mpc_rollout(bs.h, steps, None)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        n = bs.batch * K
        # compulsory bytes of one closed-loop step when the workspace lives in HBM between launches
        E_x, E_u = prob.nx * prob.N, prob.nu * (prob.N - 1)
        per_step = 4 * (2 * prob.nx + prob.nu) + 4 + (4 * (E_x + E_u) + 24 + 8 * (3 * E_u + 2 * E_x) + 16) / steps
        res[label] = {"mpc_steps_per_sec": n / dt, "ms_per_launch": 1e3 * dt / launches,
                      "algorithmic_GBps": per_step * n / dt / 1e9}
    return res


def run_sharded_capi(args):
    """`--sharded-capi`: the batch over `--gpus` devices of one node from ONE host process, through tinympc_create_sharded
    (include/tinympc_hip.h section 3) — per-shard solvers and streams, inputs scattered / outputs gathered by offset, the
    status block all-reduced over RCCL (host fold when shards share a device).  A step = one tinympc_sharded_solve: enqueue
    on every shard, wait for all, fold the status.  Same JSON line as the one-process-per-GPU path, plus the per-shard
    kernel times and the fold backend."""
    import numpy as np
    import tinympc_julia_amd as t
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit(f"bench.py: --devices names {len(devices)} shards but --gpus is {args.gpus}")
    name = args.config or ("quadrotor" if args.scaling == "strong" else "cartpole")
    tol = float(args.tol)
    check = args.check_termination or (10 if tol > 0 else 1)
    n = len(devices)
    total = (args.batch or STRONG_TOTAL) if args.scaling == "strong" else (args.batch or default_batch(name)) * n
    seed = 3 if args.scaling == "strong" else {"cartpole": 0, "quadrotor": 1}.get(name, 2)
    prob, x0, refs, label = make_workload(t, name, total, seed)
    sh = t.ShardedBatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=total, devices=devices)
    sh.update_settings(abs_pri_tol=tol, abs_dua_tol=tol, max_iter=args.iters, check_termination=check)
    sh.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    sh.set_precision(args.precision)
    sh.set_warm_start(False)
    if args.compaction > 0:
        sh.set_compaction(args.compaction)
    sh.set_x0(x0)
    if refs is not None:
        sh.set_x_ref(refs[0])
        sh.set_u_ref(refs[1])
    handles = [sh.shard(i)[3] for i in range(n)]
    for h in handles:
        sh.lib.tinympc_set_profiling(h, 1)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.15:      # clock pre-warm (set-up, not a step)
        sh.solve()
    for _ in range(args.warmup):
        sh.solve()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        status = sh.solve()                          # synchronous: every shard done, status folded
    elapsed = time.perf_counter() - t0
    k_ms = [float(sh.lib.tinympc_kernel_elapsed_mean_ms(h, args.steps)) for h in handles]
    st = sh.get_status()
    if tol <= 0.0:
        assert int(st["iter"].min()) == args.iters == int(st["iter"].max()), "work skipped"
    res, unsolved = sh.global_status()
    it_mean = float(np.mean(st["iter"]))
    ms = 1e3 * elapsed / args.steps
    how = (f"tol={tol:g}, check every {check}, max_iter={args.iters}" if tol > 0 else f"fixed {args.iters} ADMM iters")
    kmax = max(k_ms)
    per_shard = [sh.shard(i)[2] - sh.shard(i)[1] for i in range(n)]
    one = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=1, device=devices[0])   # (the formulas' carrier)
    alg_bytes = one.algorithmic_bytes() * max(per_shard)
    alg_flops = one.algorithmic_flops(1) * it_mean * max(per_shard)
    one.close()
    peak = FP64_PEAK_TFLOPS if sh.kernel_names()[0].startswith("mfma") else FP32_PEAK_TFLOPS
    out = {"metric": "qp_solves_per_sec", "value": total * args.steps / elapsed, "unit": "solves/s", "n_gpus": n,
           "library_sha256": LIB_HASH, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": "f32" if args.precision == 1 else "f32 (f64 recurrences)",
           "data": "synthetic", "mode": "sharded_capi",
           "config": {"workload": f"{label}, batch={total} in {n} contiguous shards of {per_shard[0]}, {how}, cold start",
                      "family": name, "batch_per_gpu": per_shard[0], "batch_total": total, "admm_iters_per_solve": args.iters,
                      "kernel": sh.kernel_names()[0], "devices": devices,
                      "sharding": f"ONE process, tinympc_create_sharded over {n} shard(s); status fold by {sh.fold_backend} after every solve"},
           "fold_backend": sh.fold_backend, "shard_kernel_ms": {"min": min(k_ms), "max": kmax, "all": k_ms},
           "host_overhead_ms_per_step": ms - kmax if len(set(devices)) == n else None,
           "admm_iters_per_sec": float(np.sum(st["iter"], dtype=np.float64)) * args.steps / elapsed,
           "solve_status": status, "mean_iters": it_mean, "global_residual_maxima": [float(v) for v in res],
           "roofline": {"bound": "mfma" if peak == FP64_PEAK_TFLOPS else "valu_fp64", "achieved": alg_flops / (kmax * 1e-3) / 1e12,
                        "peak": peak, "unit": "TFLOP/s", "frac": alg_flops / (kmax * 1e-3) / 1e12 / peak, "kernel_ms": kmax,
                        "hbm": {"achieved": alg_bytes / (kmax * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": alg_bytes / (kmax * 1e-3) / 1e9 / HBM_PEAK_GBS},
                        "traffic": None, "note": "per device: the slowest shard's kernel (HIP events on the shard's stream) against one GPU's peak"}}
    if len(set(devices)) != n:
        out["note"] = "shards share a device (rehearsal of the multi-shard path on fewer GPUs): they run one after the other"
    sh.close()
    print(json.dumps(out), flush=True)


# ----------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.specialised_only:
        import torch
        import tinympc_julia_amd as t
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        print(json.dumps(time_specialised(t, torch, dev, torch.cuda.Stream(device=dev))), flush=True)
        return 0
    if args.sharded_capi:
        if args.dry:
            raise SystemExit("--sharded-capi has no dry mode: rehearse it on one GPU with --gpus 2 --devices 0,0")
        return run_sharded_capi(args)
    if args.backend == "gloo" and not args.dry:
        raise SystemExit("--backend gloo is the CPU rehearsal: use it with --dry (the solver has no CPU path)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))          # before torch / HIP are even imported

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import numpy as np
    import torch
    import torch.distributed as dist

    from tinympc_julia_amd import sharding

    name = args.config or ("quadrotor" if args.scaling == "strong" else "cartpole")
    tol = float(args.tol)
    check = args.check_termination or (10 if tol > 0 else 1)
    if args.scaling == "strong":
        total = args.batch or STRONG_TOTAL
        lo, hi = sharding.shard_range(total, rank, world)
        seed = 3
    else:
        per_gpu = args.batch or default_batch(name)
        total, lo, hi = per_gpu * world, 0, per_gpu
        seed = {"cartpole": 0, "quadrotor": 1}.get(name, 2) + 1000 * rank
    n_local = hi - lo
    glo, ghi = (lo, hi) if args.scaling == "strong" else (rank * n_local, (rank + 1) * n_local)  # global instance range

    if args.dry:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # TINYMPC_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, status all-reduce) on one rank too —
    # a rehearsal of the N > 1 run on a one-GPU box
    dist_on = world > 1 or bool(os.environ.get("TINYMPC_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout is reserved for the one JSON
        # line, so the banner is sent to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.dry:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)  # RCCL on ROCm
            probe = torch.zeros(8, dtype=torch.int32, device=dev)
            dist.all_reduce(probe, op=dist.ReduceOp.MAX)      # brings the communicator up
            if not args.dry:
                torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        assert dist.get_world_size() == args.gpus, "process group size differs from --gpus"

    def sync():
        if not args.dry:
            torch.cuda.synchronize(dev)

    if args.dry:
        # no solver: the rank's "status block" is a CPU tensor carrying its shard bounds, so that the fold below
        # proves every rank took part (max hi == total) — the launch / sharding / exchange path, nothing else
        t = bs = prob = x0 = refs = None
        label = f"{name} (dry run: no solves)"
        gstat = torch.zeros(8, dtype=torch.int32)
        stream = None
    else:
        import tinympc_julia_amd as t
        if args.scaling == "strong":
            prob, x0, refs, label = make_workload(t, name, total, seed, lo, hi)
        else:
            prob, x0, refs, label = make_workload(t, name, n_local, seed)
        bs = build_solver(t, name, prob, x0, refs, local_rank, args.iters, tol, check, args.precision, args.compaction, args.keep_workspace)
        stream = torch.cuda.current_stream(dev)
        gstat = sharding.device_tensor(bs.device_buffers()["gstat"], (8,), torch.int32, dev)

    every_step = dist_on and (args.status_every_step or tol > 0)
    pending = []

    def step():
        if args.dry:
            gstat[5], gstat[6] = glo, ghi
        else:
            bs.solve_async(stream.cuda_stream)
        if every_step:
            # global max residuals / unsolved count (16 + 4 bytes) after every solve: started on a snapshot of the
            # status block and awaited at the fence, so it overlaps the next solve
            pending.append(sharding.allreduce_status_async(gstat, force=True))

    def fence():
        for _, work in pending:
            if work is not None:
                work.wait()
        del pending[:-1]          # the last one carries the global status of the last solve
        sync()
        if dist_on:
            dist.barrier()
        sync()

    if not args.dry:
        # clock pre-warm (set-up, not a step): the first milliseconds after idle run ~5 % slow while the clocks ramp
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.15:
            bs.solve_async(stream.cuda_stream)
            sync()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if dist_on and not every_step:
        # the path's only exchange in the fixed-iteration workload: one fold of the last solve's status block
        pending.append(sharding.allreduce_status_async(gstat, force=True))
        fence()

    if args.dry:
        k_ms, status, it_mean, iters_total = -1.0, 0, 0.0, 0.0
        folded = pending[-1][0].numpy() if pending else gstat.numpy()
        assert int(folded[6]) == total, "status fold did not reach every rank"
    else:
        # per-launch kernel duration: HIP events the library records on the launch stream immediately around each kernel
        k_ms = bs.kernel_elapsed_ms(args.steps)
        status = bs.solve_status()
        if dist_on and pending:
            status = max(status, sharding.decode_status(pending[-1][0].cpu().numpy())[0])
        st = bs.get_status()
        if tol <= 0.0:
            assert int(st["iter"].min()) == args.iters and int(st["iter"].max()) == args.iters, "work skipped"
        it_mean = float(np.mean(st["iter"]))
        iters_total = float(np.sum(st["iter"], dtype=np.float64))
    if dist_on:
        # ADMM iterations executed per step over all ranks (tolerance-terminated runs: data-dependent)
        tsum = torch.tensor([iters_total], device=dev, dtype=torch.float64)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        iters_total = float(tsum.item())

    if rank == 0:
        value = total * args.steps / elapsed
        avg_step_ms = 1e3 * elapsed / args.steps
        how = (f"tol={tol:g}, check every {check}, max_iter={args.iters}" if tol > 0 else f"fixed {args.iters} ADMM iters")
        shard = (f"batch={total} in total, contiguous shards of {n_local}" if args.scaling == "strong"
                 else f"batch={n_local}/GPU")
        out = {
            "metric": "qp_solves_per_sec", "value": value, "unit": "solves/s", "n_gpus": world,
            "library_sha256": LIB_HASH,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": avg_step_ms,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            # (precision 1 asks for fp32 recurrences; shapes with a matrix-core kernel run fp64 recurrences all the same — faster there)
            "dtype": ("f32" if args.precision == 1 and not (not args.dry and bs.kernel_name.startswith("mfma")) else
                      ("f64 recurrences and state trajectory, f32 input slack / dual" if (not args.dry and bs.last_launch_name.startswith("lean"))
                       else "f32 (f64 recurrences)")),
            "data": "synthetic",
            "config": {"workload": f"{label}, {shard}, {how}, cold start", "family": name,
                       "batch_per_gpu": n_local, "batch_total": total, "admm_iters_per_solve": args.iters,
                       "kernel": None if args.dry else bs.last_launch_name,
                       "kernel_family": None if args.dry else bs.kernel_name,
                       "sharding": f"batch-sharded x{world}, no data-path collective; status all-reduce "
                                   + ("every step" if every_step else "once") + (" (RCCL)" if not args.dry else " (gloo, dry)")},
            "admm_iters_per_sec": iters_total * args.steps / elapsed,
            "solve_status": status,
        }
        if args.dry:
            out["dry"] = True
            out["roofline"] = None
        else:
            kk = k_ms if k_ms > 0 else avg_step_ms
            if args.compaction > 0 and tol > 0:
                kk = avg_step_ms      # several launches per solve
            pat = pattern_of(tol, args.keep_workspace)
            out["roofline"], out["valu"] = roofline_of(bs, name, args.precision, n_local, it_mean, kk, pat)
            out["config"]["pattern"] = pat
            out["mean_iters"] = it_mean
    if not args.dry and rank == 0 and world == 1 and not dist_on:
        if args.mode == "mpc":
            out["mpc_closed_loop"] = run_mpc_mode(args, bs, prob, x0, dev, torch)
        if not args.no_cpu_baseline and name != "rocket_soc" and tol <= 0:
            out["cpu_baseline"] = cpu_baseline(prob, x0, refs, args.iters, args.cpu_seconds)
    if bs is not None:
        bs.close()
    if not args.dry and rank == 0 and world == 1 and not dist_on and not args.no_extras and args.config is None \
            and args.scaling == "weak" and tol <= 0 and args.mode == "solve":
        # BASELINE configs[2], [3] and one rank's shard of configs[4]; a few hundred ms in total
        ex = {}
        # the headline workload with the work its benched variant elides switched on: the termination check live at every
        # iteration (tolerances of 1e-30: nobody converges, every residual is formed every iteration), and a finite state bound
        # (state clamp + state dual live; runs on the quad kernel)
        ex["cartpole_65536_check_every_iteration"] = time_config(t, torch, dev, stream, "cartpole", 65536, 0, tol=1e-30)
        ex["cartpole_65536_state_bound"] = time_config(t, torch, dev, stream, "cartpole", 65536, 0, state_bound=0.45)
        # ... and in the reference's default calling pattern: the workspace kept between solves (admm.cpp:111-115), 100 fixed
        # iterations and warm-started to tolerance 1e-3 with a check every 10 (steady state: 10 iterations per solve)
        # (eight untimed solves first: from a zero workspace the warm-started solves need 68, 57, 54, 48, 23 iterations before
        # they settle at 10)
        ex["cartpole_65536_workspace_kept"] = time_config(t, torch, dev, stream, "cartpole", 65536, 0, keep_workspace=True, warmup=8)
        ex["cartpole_65536_workspace_kept_tol"] = time_config(t, torch, dev, stream, "cartpole", 65536, 0, tol=1e-3, check=10, keep_workspace=True,
                                                              warmup=8)
        ex["quadrotor_65536"] = time_config(t, torch, dev, stream, "quadrotor", 65536, 1)
        ex["rocket_soc_32768"] = time_config(t, torch, dev, stream, "rocket_soc", 32768, 2)
        ex["rocket_soc_32768_workspace_kept"] = time_config(t, torch, dev, stream, "rocket_soc", 32768, 2, keep_workspace=True)
        ex["rocket_soc_32768_check_every_iteration"] = time_config(t, torch, dev, stream, "rocket_soc", 32768, 2, tol=1e-30)
        ex["quadrotor_131072_tol"] = time_config(t, torch, dev, stream, "quadrotor", 131072, 3, tol=1e-3, check=10)
        ex["quadrotor_131072_tol_compaction"] = time_config(t, torch, dev, stream, "quadrotor", 131072, 3, tol=1e-3,
                                                            check=10, compaction=20)
        # the headline workload with the reference's adaptive rho switched on
        ex["cartpole_65536_adaptive_rho"] = time_config(t, torch, dev, stream, "cartpole", 65536, 0, adaptive=True)
        # ... and the shape the reference's adaptive rho is built for (its tables are the quadrotor's, tiny_api.cpp:269-329)
        ex["quadrotor_65536_adaptive_rho"] = time_config(t, torch, dev, stream, "quadrotor", 65536, 1, adaptive=True)
        out["configs"] = ex
        if not os.environ.get("TINYMPC_HIP_NO_JIT"):
            # in a child process: kernels compiled minutes ago on this very box must not be able to take the headline line
            # down with them (a child is not an exec of this GPU-initialised process)
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--specialised-only"], capture_output=True, text=True, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                out["specialised"] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:
                out["specialised"] = {"error": str(e)[:300]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
