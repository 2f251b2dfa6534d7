#!/usr/bin/env python3
"""Benchmark of the batched TinyMPC ADMM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one batched solve of the workload on every rank: BASELINE.json configs[1] —
cartpole nx=4 nu=1 N=20, u in [-0.5, 0.5], batch 65 536 per GPU, cold start, exactly 100 ADMM
iterations per instance (tolerances 0).  Inputs are resident in HBM before the timed region.
One process per GPU; the batch shards with no data-path collective.  The path's only exchange is an
all-reduce(MAX) over RCCL of the 5-word status block (4 residual maxima + unsolved count) that decides the
global solve status: the fixed-iteration workload needs it once, after the last solve;
--status-every-step does it after every solve, overlapped with the next one.  scaling = weak (per-GPU batch
fixed).  Rank 0 prints ONE JSON line.  Before the W warm-up steps the GPU clocks are ramped with 150 ms of untimed
solves (set-up, like building the solver); the timed region is exactly K steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector
FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD MI355X spec: dense fp64 matrix (= fp64 vector) rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cartpole", choices=["cartpole", "quadrotor", "rocket", "rocket_soc"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--precision", type=int, default=0, help="0: fp64 recurrences (default), 1: all fp32")
    ap.add_argument("--mode", default="solve", choices=["solve", "mpc"],
                    help="solve: BASELINE configs[1] (default).  mpc: warm-started closed loop (SURVEY 8f), extra")
    ap.add_argument("--mpc-steps", type=int, default=50)
    ap.add_argument("--mpc-max-iter", type=int, default=10)
    ap.add_argument("--status-every-step", action="store_true",
                    help="N > 1: all-reduce the status block after every solve (tolerance-terminated use); the default "
                         "fixed-iteration workload needs no collective (SURVEY 8e) and folds the status once at the end")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_workload(t, name, batch, rank):
    P = t.problems
    if name == "cartpole":
        prob = P.cartpole(20, u_bound=0.5)
        x0 = P.cartpole_x0(batch, seed=0 + 1000 * rank)
        refs = None
        label = "cartpole nx=4 nu=1 N=20 box-only, batch=65536/GPU, fixed 100 ADMM iters, cold start"
    elif name == "quadrotor":
        prob = P.quadrotor(30, u_bound=0.5)
        x0 = P.quadrotor_x0(batch, seed=1 + 1000 * rank)
        refs = None
        label = "quadrotor nx=12 nu=4 N=30 box, batch=65536/GPU, fixed 100 ADMM iters, cold start"
    else:
        prob = P.rocket(50)
        x0 = P.rocket_x0(batch, seed=2 + 1000 * rank)
        refs = P.rocket_refs(50)
        label = "rocket nx=6 nu=3 N=50 box-only sub-problem (no fdyn/SOC), batch=32768/GPU, fixed 100 iters"
        if name == "rocket_soc":
            label = "rocket nx=6 nu=3 N=50 SOC thrust/glide cones + box + fdyn (parity unpinned), batch=32768/GPU, fixed 100 iters"
    return prob, x0, refs, label


def cpu_baseline(prob, x0, refs, iters, seconds):
    """The reference's own compiled snapshot (oracle/_ref) when its prebuilt .so is present, else our
    C restatement, threaded over the host cores, on a bounded sample of the same workload."""
    from oracle import cpu_oracle
    kind = "ref" if cpu_oracle.have_ref() else "orc64"
    if kind == "orc64" and not os.path.isfile(cpu_oracle.PORT_LIB):
        cpu_oracle.build(port=True, ref=False)
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    xr, ur = refs if refs is not None else (None, None)
    n = min(x0.shape[1], 2048 * cores)
    done, spent = 0, 0.0
    while spent < seconds and done < 64 * x0.shape[1]:
        r = cpu_oracle.solve_batch(kind, prob, x0[:, :n], xref=xr, uref=ur, abs_pri_tol=0.0, abs_dua_tol=0.0,
                                   max_iter=iters, nthreads=cores, want_outputs=False)
        done += n
        spent += r["seconds"]
    return {"value": done / spent, "unit": "solves/s", "cores": cores,
            "kind": "reference" if kind == "ref" else "port",
            "sample": f"{done} cold-start solves of the same workload ({iters} fixed iters) in {spent:.1f} s on "
                      f"{cores} threads" + (" (compiled reference snapshot, oracle/_ref)" if kind == "ref"
                                            else " (fp64 C restatement, oracle/)")}


def streamed_state_bytes(prob, batch, iters, cones, state_bounded=True):
    """HBM bytes one launch of the run-time-horizon (stream) kernel moves by design: the per-instance trajectories
    do not fit on chip, so each ADMM iteration streams them through HBM once (stream kernel).  One-shot
    solve at fixed iterations (no residual check before the last): per knot and iteration the state-shaped arrays
    cost 3 float transfers per row and set (dual in/out + the fused backward array) and the input-shaped ones 4
    (d in/out on top), with one extra set each when cones are active."""
    sets = 2 if cones else 1
    # without a finite state bound the box set's state dual is identically zero and does not travel
    per_knot = 4.0 * ((2 * sets + 2 - (0 if state_bounded else 2)) * prob.nx + (2 * sets + 4) * prob.nu)
    return per_knot * prob.N * iters * batch


def measured_traffic(family, precision, batch, kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json): FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this same command; FETCH_SIZE doubled as the gfx950
    guide prescribes.  None when no matching measurement is committed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.isfile(path):
        return None
    for e in json.load(open(path)):
        if (e["family"], e["precision"], e["batch"], e["kernel"]) == (family, precision, batch, kernel):
            return e["hbm_bytes_per_launch"]
    return None


def measured_valu_issue(family, batch, kernel):
    """Fraction of the VALU issue cycles the kernel used, from the committed rocprofv3 SQ counter passes
    (profiles/*_sq_counters.json: SQ_INSTS_VALU x 4 cycles / (busy cycles x 1 024 SIMDs)); None if not measured."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{family}_sq_counters.json"))):
        e = json.load(open(path))
        if (e.get("kernel"), e.get("batch")) == (kernel, batch):
            return e.get("derived_valu_issue_utilisation")
    return None


def measured_mfma_issue(family, batch, kernel):
    """Fraction of the matrix cores' issue cycles the kernel used (committed rocprofv3 pass: SQ_INSTS_MFMA x 64 cycles
    over busy cycles x 1 024 SIMDs); None if not measured."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{family}_sq_counters.json"))):
        e = json.load(open(path))
        if (e.get("kernel"), e.get("batch")) == (kernel, batch):
            return e.get("derived_mfma_issue_utilisation")
    return None


def run_mpc_mode(args, t, bs, prob, x0, dev, stream, torch):
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist

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
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)  # RCCL on ROCm
            probe = torch.zeros(8, dtype=torch.int32, device=dev)
            dist.all_reduce(probe, op=dist.ReduceOp.MAX)      # brings the communicator up
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import tinympc_julia_amd as t
    from tinympc_julia_amd import sharding

    batch = args.batch or (32768 if args.config.startswith("rocket") else 65536)
    prob, x0, refs, label = make_workload(t, args.config, batch, rank)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=batch, device=local_rank)
    bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=args.iters, check_termination=1)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_precision(args.precision)
    if args.config == "rocket_soc":
        bs.set_fdyn(prob.fdyn)
        bs.set_cone_constraints([0], [3], [prob.extra["cone_mu_u"]], [0], [3], [prob.extra["cone_mu_x"]])
    bs.set_warm_start(False)          # cold start, no state I/O: compulsory traffic only
    bs.set_x0(x0)                     # H2D once; inputs stay resident in HBM
    if refs is not None:
        bs.set_x_ref(refs[0])
        bs.set_u_ref(refs[1])
    bs.set_profiling(True)
    stream = torch.cuda.current_stream(dev)
    gstat = sharding.device_tensor(bs.device_buffers()["gstat"], (8,), torch.int32, dev)

    pending = []

    def step():
        bs.solve_async(stream.cuda_stream)
        if dist_on and args.status_every_step:
            # global max residuals / unsolved count (16 + 4 bytes) after every solve: started on a snapshot of the
            # status block and awaited at the fence, so it overlaps the next solve
            pending.append(sharding.allreduce_status_async(gstat, force=True))

    def fence():
        for _, work in pending:
            if work is not None:
                work.wait()
        del pending[:-1]          # the last one carries the global status of the last solve
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # clock pre-warm (set-up, not a step): the first milliseconds after idle run ~5 % slow while the clocks ramp
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.15:
        bs.solve_async(stream.cuda_stream)
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = []
    for i in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # per-launch kernel duration: HIP events the library records on the launch stream immediately around each kernel
    kernel_ms.append(bs.kernel_elapsed_ms(args.steps))  # mean over the timed launches, events immediately around the kernel
    if dist_on:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    status = bs.solve_status()
    if dist_on and not pending:
        # the path's only exchange in the fixed-iteration workload: one fold of the last solve's status block
        pending.append(sharding.allreduce_status_async(gstat, force=True))
        fence()
    if dist_on and pending:
        status = max(status, sharding.decode_status(pending[-1][0].cpu().numpy())[0])
    st = bs.get_status()
    assert int(st["iter"].min()) == args.iters and int(st["iter"].max()) == args.iters, "work skipped"

    if rank == 0:
        total = world * batch * args.steps
        value = total / elapsed
        avg_step_ms = 1e3 * elapsed / args.steps
        k_ms = kernel_ms[-1] if kernel_ms[-1] > 0 else avg_step_ms
        alg_bytes = bs.algorithmic_bytes()          # per launch (one rank's batch)
        alg_flops = bs.algorithmic_flops(args.iters)
        ach_gbs = alg_bytes / (k_ms * 1e-3) / 1e9
        ach_tf = alg_flops / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "qp_solves_per_sec", "value": value, "unit": "solves/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == 1 else "f32 (f64 recurrences)", "data": "synthetic",
            "config": {"workload": label, "family": args.config, "batch_per_gpu": batch,
                       "admm_iters_per_solve": args.iters, "kernel": bs.kernel_name,
                       "sharding": f"batch-sharded x{world}, no data-path collective; status all-reduce " + ("every step" if args.status_every_step else "once")},
            "admm_iters_per_sec": value * args.iters,
            "solve_status": status,
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS,
                         "traffic": measured_traffic(args.config, args.precision, batch, bs.kernel_name),
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "compute-bound path (SURVEY 8d): ~430 FLOP/B; see valu"},
            "valu": {"achieved_tflops": ach_tf, "peak_tflops": FP32_PEAK_TFLOPS,
                     "frac": ach_tf / FP32_PEAK_TFLOPS, "algorithmic_flops_per_launch": alg_flops,
                     "issue_utilisation": measured_valu_issue(args.config, batch, bs.kernel_name) if args.precision == 0 else None},
        }
        if bs.kernel_name.startswith("mfma"):
            # matrix-core kernel: its roofline is the dense fp64 MFMA rate (AMD MI355X spec 78.6 TFLOP/s = half the F32
            # MFMA rate of MI355X_MICROARCH.md's table, i.e. 32 FLOP/clk/SIMD; the guide has no fp64 row of its own)
            out["roofline"].update({"bound": "mfma", "achieved": ach_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": ach_tf / FP64_MFMA_PEAK_TFLOPS,
                                    "issue_utilisation": measured_mfma_issue(args.config, batch, bs.kernel_name),
                                    "note": "algorithmic FLOPs (SURVEY 8d) over the fp64 matrix-core peak; tiles are "
                                            "padded, so the issued MFMA FLOPs are higher (DESIGN.md, mfma kernel)"})
        if bs.kernel_name.startswith("stream"):
            # the state lives in HBM by design on this kernel: its roofline is that stream, not the I/O bytes
            bounded = bool((np.asarray(prob.x_min) > -1e17).any() or (np.asarray(prob.x_max) < 1e17).any())
            sb = streamed_state_bytes(prob, batch, args.iters, args.config == "rocket_soc",
                                      bounded or args.config == "rocket_soc") + alg_bytes
            out["roofline"].update({"achieved": sb / (k_ms * 1e-3) / 1e9, "frac": sb / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "algorithmic_bytes_per_launch": sb,
                                    "note": "stream kernel: per-iteration state traffic (does not fit on chip) + I/O"})
        if args.mode == "mpc" and world == 1:
            out["mpc_closed_loop"] = run_mpc_mode(args, t, bs, prob, x0, dev, stream, torch)
        if world == 1 and not args.no_cpu_baseline and args.config != "rocket_soc":
            out["cpu_baseline"] = cpu_baseline(prob, x0, refs, args.iters, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    bs.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
