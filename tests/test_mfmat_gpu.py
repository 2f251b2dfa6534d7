"""The transposed-sets matrix-core kernel (csrc/admm_mfmat.hip.h, "mfmat<6,3,N>") — what BASELINE config 4 and the
reference's own rocket example (examples/rocket_landing_constraints.jl) run on since round 3: every kind of solve with the
affine dynamics term and one second-order cone per side, the whole iterated state on chip.
 * one-shot solves against the fp64 oracle, every instance by solution (tests/util.parity_every_instance), and against the
   three-wavefront kernel of round 2 it replaces (bit for bit at N = 10);
 * the reference's DEFAULT calling pattern — the workspace persists between solves (admm.cpp:111-115) — as a host-stepped
   closed loop with the workspace itself compared after every solve, converged exits included (the "v, z, d one
   iteration old" quirk of admm.cpp:181-197);
 * the fused closed loop with the references shifted every step and the affine plant step
   (rocket_landing_constraints.jl:97-134) against the oracle driven step by step on the host;
 * chunked solves with compaction, BASELINE config 4 at its full size with 32 768 DISTINCT instances.
Cones / fdyn are the UNPINNED extensions (no reference source): the oracle itself is pinned for them by
tests/test_independent_optimum.py and tests/test_extensions_cpu.py."""
import numpy as np
import pytest

import tinympc_julia_amd as t
from tests.util import FP32_TOL, nrel, nrel_batch, parity_every_instance

pytestmark = pytest.mark.gpu

ROCKET_CONES = ([0], [3], [0.25], [0], [3], [0.5])          # inputs first (bindings.cpp:453-459)
SETTINGS = {
    "fixed60": dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=1),
    "tol": dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1),   # rocket_landing_constraints.jl:61-62
    "tol_ct10": dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=95, check_termination=10),
    "fixed_ct7": dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=45, check_termination=7),   # last check at 42, three more iterations
    "nocheck": dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=30, check_termination=0),
    "one_iter": dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=1, check_termination=1),
}


def _configure(o, prob, kw, xr, ur, fdyn, cones):
    o.update_settings(**kw)
    o.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if fdyn is not None:
        o.set_fdyn(fdyn)
    if cones is not None:
        o.set_cone_constraints(*cones)
    if xr is not None:
        o.set_x_ref(xr)
        o.set_u_ref(ur)
    return o


def _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones):
    def make(b=None):
        return _configure(oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N), prob, kw, xr, ur,
                          fdyn, cones)
    return make


def _loop(make, x0):
    B = x0.shape[1]
    out = None
    for b in range(B):
        o = make(b)
        o.set_x0(x0[:, b])
        o.solve()
        r = o.get_solution()
        if out is None:
            out = dict(x=np.zeros(r["x"].shape + (B,)), u=np.zeros(r["u"].shape + (B,)), iter=np.zeros(B, dtype=int),
                       solved=np.zeros(B, dtype=int), res=np.zeros((B, 4)))
        out["x"][:, :, b], out["u"][:, :, b] = r["x"], r["u"]
        out["iter"][b], out["solved"][b], out["res"][b] = r["iter"], r["solved"], r["res"]
        o.close()
    return out


def _solver(prob, B, kw, xr, ur, fdyn, cones, warm):
    bs = _configure(t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B), prob, kw, xr, ur, fdyn, cones)
    bs.set_warm_start(warm)
    return bs


@pytest.mark.parametrize("N", [10, 50])
@pytest.mark.parametrize("mode", ["fdyn+cones", "fdyn", "cones", "state_cone", "zero_refs"])
@pytest.mark.parametrize("setting", list(SETTINGS))
def test_mfmat_one_shot_vs_oracle(hip_lib, oracle_built, N, mode, setting):
    if mode in ("state_cone", "zero_refs") and setting not in ("fixed60", "tol"):
        pytest.skip("the single-cone / zero-reference variants run the two main settings only")
    B = 37                                                  # ragged: two full tiles of 16 and one of 5
    kw = SETTINGS[setting]
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=2)
    xr, ur = t.problems.rocket_refs(N)
    if mode == "zero_refs":
        xr = ur = None
    fdyn = prob.fdyn if mode != "cones" else None
    cones = ROCKET_CONES if "cones" in mode or mode == "zero_refs" else None
    if mode == "state_cone":
        cones = ([], [], [], [0], [3], [0.5])
    mk = _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, xr, ur, fdyn, cones, False)
    bs.set_x0(x0)
    status = bs.solve()
    assert bs.kernel_name == f"mfmat<6,3,{N}>"
    sol, st = bs.get_solution(), bs.get_status()
    assert status == int(np.any(st["solved"] == 0))
    parity_every_instance(sol, st, ref, mk, x0, kw, prob.rho, tag=f"N={N} {mode} {setting}")
    same = st["iter"] == ref["iter"]
    if kw["check_termination"] > 0:
        # residuals are differences of fp32 values of magnitude up to ~100 (thrust): a few ulp of those is the floor
        atol = 4e-7 * max(1.0, np.abs(ref["x"]).max(), np.abs(ref["u"]).max()) * max(1.0, prob.rho)
        assert np.allclose(st["residuals"][same], ref["res"][same], rtol=1e-2, atol=atol), \
            np.abs(st["residuals"][same] - ref["res"][same]).max()
    # a second solve of the same inputs returns the same bits (nothing of the first one survives)
    bs.solve()
    assert np.array_equal(bs.get_solution()["controls"], sol["controls"])
    assert np.array_equal(bs.get_status()["iter"], st["iter"])
    bs.close()


@pytest.mark.parametrize("N", [10, 50])
@pytest.mark.parametrize("setting", ["fixed60", "tol"])
def test_mfmat_against_three_wavefront_kernel(hip_lib, monkeypatch, N, setting):
    """same solve on the kernel of round 2 (one knot at a time, sets in the matrix layout, three products per step): the same
    fp32 state arithmetic in another lane layout, the fp64 recurrences summed in another order"""
    B = 70
    kw = SETTINGS[setting]
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=12)
    xr, ur = t.problems.rocket_refs(N)
    outs = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("TINYMPC_HIP_NO_MFMAT", env)
        bs = _solver(prob, B, kw, xr, ur, prob.fdyn, ROCKET_CONES, False)
        bs.set_x0(x0)
        bs.solve()
        outs.append((bs.kernel_name, bs.get_solution(), bs.get_status()))
        bs.close()
    assert outs[0][0] == f"mfmat<6,3,{N}>" and outs[1][0] == f"mfmar<6,3,{N}>"
    same = outs[0][2]["iter"] == outs[1][2]["iter"]
    assert same.mean() >= 0.97
    assert nrel_batch(outs[0][1]["states"], outs[1][1]["states"])[same].max() <= 1e-6
    assert nrel_batch(outs[0][1]["controls"], outs[1][1]["controls"])[same].max() <= 1e-6


@pytest.mark.parametrize("N,setting,steps", [(10, "tol", 6), (10, "fixed_ct7", 3), (50, "tol", 3), (50, "fixed60", 2), (10, "tol_ct10", 4)])
def test_mfmat_workspace_persists_like_the_reference(hip_lib, oracle_built, N, setting, steps):
    """The reference's default: solve() resets counters only (admm.cpp:111-115), the workspace carries over.  A host-stepped
    closed loop (set_x0 -> solve -> x+ = A x + B u0 + f) with one persistent fp64 oracle per instance beside it: solution,
    iteration count and the WORKSPACE ITSELF (d, y, g, v, z) after every solve — after a converged exit v, z and d are the
    previous iteration's (admm.cpp:181-197), which is what the next solve's first residual check and rollout see."""
    B = 40
    kw = SETTINGS[setting]
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=5)
    xr, ur = t.problems.rocket_refs(N)
    bs = _solver(prob, B, kw, xr, ur, prob.fdyn, ROCKET_CONES, True)
    orcs = [_oracle(oracle_built, prob, kw, xr, ur, prob.fdyn, ROCKET_CONES)() for _ in range(B)]
    x = x0.copy()
    converged = replayed = 0
    for k in range(steps):
        bs.set_x0(x)
        bs.solve()
        assert bs.kernel_name == f"mfmat<6,3,{N}>"
        sol, st, ws = bs.get_solution(), bs.get_status(), bs.get_workspace()
        xn = np.zeros_like(x)
        for b in range(B):
            o = orcs[b]
            o.set_x0(x[:, b])
            pre, pre_c = o.get_state(), o.get_cone_state()
            o.solve()
            r = o.get_solution()
            assert abs(int(st["iter"][b]) - r["iter"]) <= max(1, kw["check_termination"])
            if int(st["iter"][b]) != r["iter"]:
                # a residual within rounding of the tolerance: the oracle repeats the solve from the same workspace with
                # the GPU's termination decision imposed, and is compared like every other instance
                o.set_state(*[pre[key] for key in ("d", "y", "g", "v", "z")])
                o.set_cone_state(**pre_c)
                o.set_forced_exit(int(st["iter"][b]) if st["solved"][b] else -1)
                o.solve()
                o.set_forced_exit(0)
                r = o.get_solution()
                assert r["iter"] == st["iter"][b]
                replayed += 1
            sv = o.get_state()
            converged += r["solved"]
            ex_, eu_ = nrel(sol["states"][:, :, b], r["x"]), nrel(sol["controls"][:, :, b], r["u"])
            assert ex_ <= FP32_TOL and eu_ <= FP32_TOL, f"step {k} instance {b}: x {ex_:.3e} u {eu_:.3e}"
            for key in ("d", "y", "g", "v", "z"):
                scale = max(np.abs(sv[key]).max(), 1e-2)
                e_ = np.abs(ws[key][:, :, b] - sv[key]).max() / scale
                lim = 2e-5 if key in ("g", "y") else FP32_TOL   # the duals integrate the trajectory's per-iteration rounding
                assert e_ <= lim, f"step {k} instance {b} workspace {key}: {e_:.3e}"
            xn[:, b] = prob.A @ x[:, b] + prob.B @ r["u"][:, 0] + prob.fdyn
        x = xn
    if setting.startswith("tol"):
        assert converged >= (B if N == 10 else 10)         # the converged-exit path is exercised
    assert replayed <= 0.05 * B * steps
    for o in orcs:
        o.close()
    bs.close()


def _rocket_ref_sequence(N, steps, ntotal=100):
    """x_ref of step k (1-based), as rocket_landing_constraints.jl:107-115 sets it before each solve"""
    xinit = np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5])
    xgoal = np.zeros(6)
    xs = np.zeros((6, N, steps))
    us = np.zeros((3, N - 1, steps))
    for k in range(1, steps + 1):
        for i in range(1, N + 1):
            xs[:, i - 1, k - 1] = xinit + (xgoal - xinit) * (i + k - 2) / (ntotal - 1)
        us[2, :, k - 1] = 10.0
    return xs, us


@pytest.mark.parametrize("shifted", [True, False])
def test_mfmat_fused_rocket_loop_vs_oracle(hip_lib, oracle_built, shifted):
    """The reference's rocket example as ONE launch: examples/rocket_landing_constraints.jl:97-134 — set_x0, references
    shifted by one knot (:107-115), solve (warm-started, tolerances 2e-3 / 1e-3, max_iter 100), x+ = A x + B u0 + f (:123) —
    for a batch of perturbed initial states, against the fp64 oracle driven step by step on the host.  Steps where the
    fp32 residual falls on the other side of the tolerance are replayed on the oracle with the GPU's decision imposed."""
    N, B, steps = 10, 48, 24
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=21)
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    xs, us = _rocket_ref_sequence(N, steps)
    if not shifted:
        xs, us = np.repeat(xs[:, :, :1], steps, axis=2), np.repeat(us[:, :, :1], steps, axis=2)
    ref_u, ref_x, ref_it = np.zeros((3, steps, B)), np.zeros((6, steps, B)), np.zeros((steps, B), dtype=int)

    def oracle_loop(b, forced=None):
        o = _configure(oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N), prob, kw, xs[:, :, 0],
                       us[:, :, 0], prob.fdyn, ROCKET_CONES)
        x = x0[:, b].copy()
        for k in range(steps):
            if forced is not None:
                o.set_forced_exit(int(forced[k][0]) if forced[k][1] else -1)
            o.set_x0(x)
            o.set_x_ref(xs[:, :, k])
            o.set_u_ref(us[:, :, k])
            o.solve()
            r = o.get_solution()
            x = prob.A @ x + prob.B @ r["u"][:, 0] + prob.fdyn
            ref_u[:, k, b], ref_x[:, k, b], ref_it[k, b] = r["u"][:, 0], x, r["iter"]
        o.close()

    for b in range(B):
        oracle_loop(b)
    bs = _solver(prob, B, kw, xs[:, :, 0], us[:, :, 0], prob.fdyn, ROCKET_CONES, True)
    if shifted:
        bs.set_ref_sequence(xs, us)
    bs.set_x0(x0)
    log = bs.mpc_rollout(steps)
    assert bs.kernel_name == "mfmat<6,3,10>"
    same = np.all(log["iter"] == ref_it, axis=0)
    assert same.mean() >= 0.85, same.mean()
    for b in np.nonzero(~same)[0]:
        assert np.abs(log["iter"][:, b] - ref_it[:, b]).max() <= 1
        oracle_loop(b, [(log["iter"][k, b], log["solved"][k, b]) for k in range(steps)])
        assert np.array_equal(ref_it[:, b], log["iter"][:, b])
    assert len(np.unique(log["iter"])) > 3 and log["solved"].mean() > 0.5     # early exits at different iterations
    eu = np.abs(log["u"] - ref_u).max(axis=(0, 1)) / np.abs(ref_u).max(axis=(0, 1))
    exx = np.abs(log["x"] - ref_x).max(axis=(0, 1)) / np.abs(ref_x).max(axis=(0, 1))
    assert eu.max() <= FP32_TOL, f"applied controls: worst {eu.max():.3e} (instance {eu.argmax()})"
    assert exx.max() <= FP32_TOL, f"plant states: worst {exx.max():.3e} (instance {exx.argmax()})"
    # the last solve's outputs and the plant state are where the separate entry points find them
    xl = bs.get_x0() if hasattr(bs, "get_x0") else None
    if xl is not None:
        assert np.abs(xl - ref_x[:, -1, :]).max() <= 1e-5 * np.abs(ref_x).max()
    # the same loop stepped by the host (what a caller without the fused entry point does): same kernel, same path
    bs2 = _solver(prob, B, kw, xs[:, :, 0], us[:, :, 0], prob.fdyn, ROCKET_CONES, True)
    x = x0.copy()
    u2, it2 = np.zeros_like(log["u"]), np.zeros_like(log["iter"])
    for k in range(steps):
        bs2.set_x0(x)
        bs2.set_x_ref(xs[:, :, k])
        bs2.set_u_ref(us[:, :, k])
        bs2.solve()
        u2[:, k, :] = bs2.get_solution()["controls"][:, 0, :]
        it2[k] = bs2.get_status()["iter"]
        x = prob.A @ x + prob.B @ u2[:, k, :] + prob.fdyn[:, None]
    agree = np.all(it2 == log["iter"], axis=0)             # the host loop rounds the plant state to fp32 every step
    assert agree.mean() >= 0.85
    e2 = np.abs(u2 - log["u"]).max(axis=(0, 1)) / np.abs(ref_u).max(axis=(0, 1))
    assert e2[agree].max() <= FP32_TOL, f"host-stepped loop vs fused loop: {e2[agree].max():.3e}"
    bs.close(); bs2.close()


def test_mfmat_fused_loop_shifted_refs_many_tiles(hip_lib, oracle_built):
    """A persistent workgroup that takes a SECOND tile must solve that tile's step 0 against step 0's references, not the
    ones the previous tile's last step left in LDS (round-3 advisor finding).  A batch far beyond what the launch has
    workgroups for, identical x0 everywhere: every instance must be bit-equal to instance 0, and instance 0 must be the
    oracle's loop (rocket_landing_constraints.jl:97-134)."""
    N, B, steps = 10, 49152, 6            # 3 072 tiles; a launch holds at most 256 CUs x 8 workgroups
    prob = t.problems.rocket(N)
    x1 = t.problems.rocket_x0(1, seed=5)
    x0 = np.repeat(x1, B, axis=1)
    kw = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=12, check_termination=1)
    xs, us = _rocket_ref_sequence(N, steps)
    bs = _solver(prob, B, kw, xs[:, :, 0], us[:, :, 0], prob.fdyn, ROCKET_CONES, True)
    bs.set_ref_sequence(xs, us)
    bs.set_x0(x0)
    log = bs.mpc_rollout(steps)
    assert bs.kernel_name == "mfmat<6,3,10>"
    for key in ("u", "x"):
        assert np.array_equal(log[key], np.repeat(log[key][:, :, :1], B, axis=2)), f"{key}: instances differ between tiles"
    o = _configure(oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N), prob, kw, xs[:, :, 0],
                   us[:, :, 0], prob.fdyn, ROCKET_CONES)
    x = x1[:, 0].copy()
    ref_u, ref_x = np.zeros((3, steps)), np.zeros((6, steps))
    for k in range(steps):
        o.set_x0(x)
        o.set_x_ref(xs[:, :, k])
        o.set_u_ref(us[:, :, k])
        o.solve()
        u0 = o.get_solution()["u"][:, 0]
        x = prob.A @ x + prob.B @ u0 + prob.fdyn
        ref_u[:, k], ref_x[:, k] = u0, x
    o.close()
    assert np.abs(log["u"][:, :, 0] - ref_u).max() <= FP32_TOL * np.abs(ref_u).max()
    assert np.abs(log["x"][:, :, -1] - ref_x).max() <= FP32_TOL * np.abs(ref_x).max()
    bs.close()


def test_mfmat_chunked_solve_with_compaction(hip_lib):
    """tolerance-terminated solve in chunks with the unconverged instances gathered between them (tinympc_set_compaction):
    the workspace carries every instance from chunk to chunk, so the result is the single launch's"""
    N, B = 10, 3000
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=8)
    x0[:, ::3] *= 0.2
    xr, ur = t.problems.rocket_refs(N)
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    outs = []
    for chunk in (0, 16):
        bs = _solver(prob, B, kw, xr, ur, prob.fdyn, ROCKET_CONES, True)
        bs.set_compaction(chunk)
        bs.set_x0(x0)
        status = bs.solve()
        assert bs.kernel_name == "mfmat<6,3,10>"
        outs.append((status, bs.get_solution(), bs.get_status(), bs.get_workspace()))
        bs.close()
    assert outs[0][0] == outs[1][0]
    assert len(np.unique(outs[0][2]["iter"])) > 5
    # an instance that converges inside a chunk leaves exactly as in the single launch; one that crosses a chunk boundary
    # re-enters through the fp32 workspace (t -> d -> t), which moves the last bits of its later iterates
    same = outs[0][2]["iter"] == outs[1][2]["iter"]
    assert same.mean() >= 0.98
    assert nrel_batch(outs[0][1]["controls"], outs[1][1]["controls"])[same].max() <= 2e-6
    assert nrel_batch(outs[0][1]["states"], outs[1][1]["states"])[same].max() <= 2e-6
    assert np.array_equal(outs[0][2]["solved"][same], outs[1][2]["solved"][same])


def test_mfmat_config4_every_instance(hip_lib, oracle_built):
    """BASELINE config 4 exactly as benchmarked — rocket N = 50, input cone + state cone + box + affine term, batch 32 768
    DISTINCT instances (seed 2), 100 fixed iterations — every instance within 1e-5 (norm-relative) of the fp64 oracle: cold
    one-shot, and with the reference's default (workspace kept), where a second solve then continues from the first one's
    state (checked on the first 64 instances against oracles that solve twice)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    B, N, D = 32768, 50, 64
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=2)
    xr, ur = t.problems.rocket_refs(N)
    kw = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
    X, U, R4 = np.zeros((6, N, B)), np.zeros((3, N - 1, B)), np.zeros((B, 4))
    X2, U2 = np.zeros((6, N, D)), np.zeros((3, N - 1, D))
    quota = os.cpu_count() or 1
    try:                                                   # the box's cgroup quota, not the visible CPU count (bench.py)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except OSError:
        pass
    workers = max(1, min(16, quota))
    mk = _oracle(oracle_built, prob, kw, xr, ur, prob.fdyn, ROCKET_CONES)

    def work(w):                                           # (ctypes releases the GIL: the workers run side by side)
        o = mk()
        for b in range(w, B, workers):
            o.reset()
            o.set_x0(x0[:, b])
            o.solve()
            r = o.get_solution()
            X[:, :, b], U[:, :, b], R4[b] = r["x"], r["u"], r["res"]
            if b < D:                                      # the reference's default: the same inputs again, workspace kept
                o.solve()
                r = o.get_solution()
                X2[:, :, b], U2[:, :, b] = r["x"], r["u"]
        o.close()

    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(work, range(workers)))
    atol = 4e-7 * max(1.0, np.abs(X).max(), np.abs(U).max())
    for warm in (False, True):
        bs = _solver(prob, B, kw, xr, ur, prob.fdyn, ROCKET_CONES, warm)
        bs.set_x0(x0)
        assert bs.solve() == 1
        assert bs.kernel_name == "mfmat<6,3,50>"
        sol, st = bs.get_solution(), bs.get_status()
        ex_, eu_ = nrel_batch(sol["states"], X), nrel_batch(sol["controls"], U)
        assert ex_.max() <= FP32_TOL and eu_.max() <= FP32_TOL, (warm, ex_.max(), eu_.max())
        assert np.all(st["iter"] == 100) and np.all(st["solved"] == 0)
        assert np.allclose(st["residuals"], R4, rtol=1e-2, atol=atol)
        if warm:
            bs.solve()
            sol = bs.get_solution()
            ex_, eu_ = nrel_batch(sol["states"][:, :, :D], X2), nrel_batch(sol["controls"][:, :, :D], U2)
            assert ex_.max() <= FP32_TOL and eu_.max() <= FP32_TOL, ("second solve", ex_.max(), eu_.max())
        bs.close()


def test_mfmat_selection(hip_lib, monkeypatch):
    """what runs where: the transposed-sets kernel takes the affine term / one cone per side of the rocket's layout in every
    calling pattern, with shared, zero or per-instance references; other cone layouts and linear rows stay where they were (fp32 recurrences only if TINYMPC_HIP_STRICT_FP32 insists)"""
    prob = t.problems.rocket(10)
    xr, ur = t.problems.rocket_refs(10)
    kw = SETTINGS["tol"]
    bs = _solver(prob, 8, kw, xr, ur, prob.fdyn, ROCKET_CONES, True)
    assert bs.kernel_name == "mfmat<6,3,10>"
    bs.set_cone_constraints([0], [3], [0.25], [1], [3], [0.5])     # state cone on rows 1..3: not the compiled layout
    assert bs.kernel_name == "stream4<6,3>"
    bs.set_cone_constraints(*ROCKET_CONES)
    assert bs.kernel_name == "mfmat<6,3,10>"
    bs.set_x0(t.problems.rocket_x0(8, seed=1))
    bs.set_precision(1)                                            # (the kernel is chosen when a solve is launched)
    bs.solve()
    assert bs.kernel_name == "mfmat<6,3,10>"                       # fp32 recurrences are asked for to save time: no saving here
    monkeypatch.setenv("TINYMPC_HIP_STRICT_FP32", "1")
    bs.reload_switches()                                           # (the environment is read once, at creation)
    bs.solve()
    assert bs.kernel_name == "stream4<6,3>"
    monkeypatch.delenv("TINYMPC_HIP_STRICT_FP32")
    bs.reload_switches()
    assert bs.effective_precision == 0                             # (what runs is reported: fp64 recurrences on the matrix cores)
    bs.set_strict_precision(True)                                  # the API form of the same request
    bs.solve()
    assert bs.kernel_name == "stream4<6,3>" and bs.effective_precision == 1
    bs.set_strict_precision(False)
    bs.set_precision(0)
    bs.solve()
    assert bs.kernel_name == "mfmat<6,3,10>"
    bs.set_x_ref(np.repeat(xr[:, :, None], 8, axis=2))             # per-instance references
    bs.set_u_ref(np.repeat(ur[:, :, None], 8, axis=2))
    bs.set_x0(t.problems.rocket_x0(8, seed=1))
    bs.solve()
    assert bs.kernel_name == "mfmat<6,3,10>"                       # (round 3: a second set of LDS cells per tile)
    bs.set_linear_constraints(np.array([[0.0, 0.0, -1.0, 0.0, 0.0, 0.3]]), [0.5], np.zeros((0, 3)), [])
    bs.solve()
    assert bs.kernel_name == "stream4<6,3>"                        # linear rows: not here
    bs.close()
    # box-only solves of the shape run here too (measured faster than the quad kernel at every compiled horizon)
    p10 = t.problems.rocket(10)
    b10 = t.BatchSolver(p10.A, p10.B, p10.Q, p10.R, p10.rho, p10.N, batch=4096)
    b10.set_bound_constraints(p10.x_min, p10.x_max, p10.u_min, p10.u_max)
    assert b10.kernel_name == "mfmat<6,3,10>"
    b10.close()


def test_fp32_host_entry_points(hip_lib):
    """the Float32 forms of the per-solve transfers (set_x0_f32 / get_states_f32 / get_controls_f32, handle and
    process-global): plain copies of the fp32 device buffers — the same bits the fp64 forms widen"""
    import ctypes
    prob = t.problems.cartpole(20, u_bound=0.5)
    B = 300
    x0 = t.problems.cartpole_x0(B, seed=3)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=40, check_termination=1)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    bs.set_x0(x0)
    bs.solve()
    ref = bs.get_solution()
    bs.reset()
    bs.set_x0_f32(x0.astype(np.float32))
    bs.solve()
    got = bs.get_solution_f32()
    assert got["states"].dtype == np.float32 and got["states"].shape == (4, 20, B)
    assert np.array_equal(got["states"].astype(np.float64), ref["states"])
    assert np.array_equal(got["controls"].astype(np.float64), ref["controls"])
    bs.set_x0_f32(x0[:, 5].astype(np.float32))            # one state broadcast to the batch
    bs.reset()
    bs.solve()
    one = bs.get_solution_f32()
    assert np.array_equal(one["controls"][:, :, 0], one["controls"][:, :, B - 1])
    assert np.array_equal(one["controls"][:, :, 0].astype(np.float64), ref["controls"][:, :, 5])
    # caller-controlled page-locking (the library pins nothing by itself: round-3 advisor finding): pinned buffers carry the
    # same bits, a second pin of the same range is a no-op, unpinning an unknown address is an error
    keep_x, keep_u = np.zeros((4, 20, B), dtype=np.float32, order="F"), np.zeros((1, 19, B), dtype=np.float32, order="F")
    bs.pin_host(keep_x); bs.pin_host(keep_u); bs.pin_host(keep_x)
    again = bs.get_solution_f32(keep_x, keep_u)
    assert again["states"] is keep_x and np.array_equal(keep_x, one["states"]) and np.array_equal(keep_u, one["controls"])
    bs.unpin_host(keep_x)
    with pytest.raises(t.TinyMPCError):
        bs.unpin_host(keep_x)
    bs.close()                                             # (keep_u is still registered: destroy releases it)
    # the process-global forms a Julia host binds
    lib = hip_lib
    s = t.TinyMPCSolver()
    t.setup(s, prob.A, prob.B, np.zeros(4), prob.Q, prob.R, prob.rho, 4, 1, 20, batch=B, max_iter=40, abs_pri_tol=0.0, abs_dua_tol=0.0)
    t.set_bound_constraints(s, prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    x32 = np.asfortranarray(x0.astype(np.float32))
    assert lib.set_x0_f32(fp(x32), 4, B, 0) == 0
    assert t.solve(s) == 1
    xs, us = np.zeros(4 * 20 * B, dtype=np.float32), np.zeros(19 * B, dtype=np.float32)
    r, c = ctypes.c_int(), ctypes.c_int()
    assert lib.get_states_f32(fp(xs), ctypes.byref(r), ctypes.byref(c)) == 0 and (r.value, c.value) == (4, 20 * B)
    assert lib.get_controls_f32(fp(us), ctypes.byref(r), ctypes.byref(c)) == 0 and (r.value, c.value) == (1, 19 * B)
    assert np.array_equal(us.reshape((1, 19, B), order="F").astype(np.float64), ref["controls"])
    assert lib.set_x0_f32(fp(x32), 3, B, 0) == -1
    assert lib.pin_host_buffer(xs.ctypes.data_as(ctypes.c_void_p), xs.nbytes) == 0
    assert lib.get_states_f32(fp(xs), ctypes.byref(r), ctypes.byref(c)) == 0
    assert np.array_equal(xs.reshape((4, 20, B), order="F").astype(np.float64), ref["states"])
    assert lib.unpin_host_buffer(xs.ctypes.data_as(ctypes.c_void_p)) == 0
    assert lib.unpin_host_buffer(xs.ctypes.data_as(ctypes.c_void_p)) == -1
    t.cleanup()


@pytest.mark.parametrize("shape", ["quadrotor_N10", "quadrotor_N25", "cartpole_N15", "cartpole_N5"])
def test_further_horizons_have_on_chip_kernels(hip_lib, oracle_built, shape):
    """horizons beyond the examples' (quadrotor 10 / 15 / 25, cartpole 5 / 15 / 30: tests/test_codegen.jl:15 uses N = 5) used to
    fall to the HBM-streaming run-time-horizon kernel; they now have matrix-core / lanes-per-instance instantiations —
    against the fp64 oracle, cold and with the workspace kept, fixed-iteration and tolerance-terminated"""
    B = 150
    if shape.startswith("quadrotor"):
        N = int(shape.split("N")[1])
        prob, x0, kernel = t.problems.quadrotor(N, u_bound=0.5), t.problems.quadrotor_x0(B, seed=41), f"mfma<12,4,{N}>"
    else:
        N = int(shape.split("N")[1])
        prob, x0, kernel = t.problems.cartpole(N, u_bound=0.5), t.problems.cartpole_x0(B, seed=41), f"quad<4,1,{N},g4>"
    for kw in (dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=50, check_termination=1),
               dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)):
        mk = _oracle(oracle_built, prob, kw, None, None, None, None)
        ref = _loop(mk, x0)
        for warm in (False, True):
            bs = _solver(prob, B, kw, None, None, None, None, warm)
            bs.set_x0(x0)
            bs.solve()
            assert bs.kernel_name == kernel, bs.kernel_name
            parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, min_same=0.95,
                                  tag=f"{shape} {kw['max_iter']} warm={warm}")
            bs.close()


def test_rocket_loop_through_dropin_api(hip_lib):
    """the fused rocket loop on the process-global entry points a Julia host binds (setup with fdyn and a batch,
    set_bound_constraints, set_cone_constraints, set_ref_sequence, set_x0, mpc_rollout): the handle API's results, bit for bit"""
    N, B, steps = 10, 64, 12
    prob = t.problems.rocket(N)
    x0 = t.problems.rocket_x0(B, seed=33)
    xs, us = _rocket_ref_sequence(N, steps)
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    bs = _solver(prob, B, kw, xs[:, :, 0], us[:, :, 0], prob.fdyn, ROCKET_CONES, True)
    bs.set_ref_sequence(xs, us)
    bs.set_x0(x0)
    ref = bs.mpc_rollout(steps)
    bs.close()
    s = t.TinyMPCSolver()
    t.setup(s, prob.A, prob.B, prob.fdyn, prob.Q, prob.R, prob.rho, 6, 3, N, batch=B, max_iter=100, abs_pri_tol=2e-3, abs_dua_tol=1e-3)
    t.set_bound_constraints(s, prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    t.set_cone_constraints(s, *ROCKET_CONES)
    t.set_ref_sequence(s, xs, us)
    t.set_x0(s, x0)
    got = t.mpc_rollout(s, steps)
    assert t.kernel_name() == "mfmat<6,3,10>"
    assert got["status"] == ref["status"]
    for k in ("x", "u", "iter", "solved"):
        assert np.array_equal(got[k], ref[k]), k
    t.cleanup()


@pytest.mark.parametrize("N", [10, 50])
@pytest.mark.parametrize("warm", [False, True])
def test_mfmat_per_knot_bounds_random_family(hip_lib, oracle_built, N, warm):
    """bounds that depend on the knot (the kernel's per-knot bound pack in LDS instead of scalars), a random stable family
    instead of the rocket, no state bounds on half the rows: one-shot and with the workspace kept (second solve from the
    plant's next state), every instance against the oracle"""
    rng = np.random.default_rng(100 + N)
    nx, nu, B = 6, 3, 45
    A = np.eye(nx) + 0.2 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= 0.96 / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)),
                              np.diag(rng.uniform(0.5, 3.0, nu)), float(rng.uniform(0.5, 2.0)), N)
    prob.x_min, prob.x_max = -rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N)), rng.uniform(0.8, 2.0, (nx, 1)) * np.ones((1, N))
    prob.x_min[3:, :], prob.x_max[3:, :] = -1e17, 1e17
    prob.x_min[:3, N // 2:] -= 0.3                          # per-knot
    prob.u_min, prob.u_max = -rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1)), rng.uniform(0.2, 0.6, (nu, 1)) * np.ones((1, N - 1))
    prob.u_max[:, ::2] += 0.1
    fdyn = 0.02 * rng.standard_normal(nx)
    xr, ur = 0.2 * rng.standard_normal((nx, N)), 0.1 * rng.standard_normal((nu, N - 1))
    cones = ([0], [3], [0.7], [0], [3], [1.1])
    kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60, check_termination=1)
    x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    mk = _oracle(oracle_built, prob, kw, xr, ur, fdyn, cones)
    bs = _solver(prob, B, kw, xr, ur, fdyn, cones, warm)
    orcs = [mk() for _ in range(B)]
    x = x0.copy()
    for k in range(2 if warm else 1):
        ref = dict(x=np.zeros((nx, N, B)), u=np.zeros((nu, N - 1, B)), iter=np.zeros(B, dtype=int), solved=np.zeros(B, dtype=int),
                   res=np.zeros((B, 4)))
        pre = []
        for b in range(B):
            o = orcs[b]
            o.set_x0(x[:, b])
            pre.append((o.get_state(), o.get_cone_state()))
            o.solve()
            r = o.get_solution()
            ref["x"][:, :, b], ref["u"][:, :, b], ref["iter"][b], ref["solved"][b], ref["res"][b] = r["x"], r["u"], r["iter"], r["solved"], r["res"]
        bs.set_x0(x)
        bs.solve()
        assert bs.kernel_name == f"mfmat<6,3,{N}>"

        def replay(b):                                       # an oracle at instance b's state BEFORE this solve
            o = mk()
            o.set_state(*[pre[b][0][key] for key in ("d", "y", "g", "v", "z")])
            o.set_cone_state(**pre[b][1])
            return o
        parity_every_instance(bs.get_solution(), bs.get_status(), ref, replay, x, kw, prob.rho, tol=2e-5 if k else FP32_TOL,
                              min_same=0.9, tag=f"N={N} warm={warm} solve {k}")
        x = prob.A @ x + prob.B @ ref["u"][:, 0, :] + fdyn[:, None]
    for o in orcs:
        o.close()
    bs.close()


@pytest.mark.parametrize("N", [10, 50])
def test_mfmat_per_instance_references(hip_lib, oracle_built, N):
    """every instance tracks its OWN reference trajectory (`set_x_ref` / `set_u_ref` with [nx, N, B] / [nu, N-1, B] arrays):
    a second set of LDS cells per tile and the terminal term per instance; one-shot with the termination check live, then
    two workspace-carrying fixed-iteration solves against persistent oracles"""
    B = 37
    rng = np.random.default_rng(31 + N)
    prob = t.problems.rocket(N)
    xr, ur = t.problems.rocket_refs(N)
    xr3 = xr[:, :, None] * (1.0 + 0.15 * rng.standard_normal((1, 1, B))) + 0.3 * rng.standard_normal((prob.A.shape[0], 1, B))
    ur3 = ur[:, :, None] + 0.5 * rng.standard_normal((prob.B.shape[1], N - 1, B))
    x0 = t.problems.rocket_x0(B, seed=4)

    def oracle(kw):
        def make(b):
            return _configure(oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N), prob, kw,
                              np.asfortranarray(xr3[:, :, b]), np.asfortranarray(ur3[:, :, b]), prob.fdyn, ROCKET_CONES)
        return make

    # --- one-shot, tolerance-terminated ---
    kw = SETTINGS["tol"]
    mk = oracle(kw)
    ref = _loop(mk, x0)
    bs = _solver(prob, B, kw, np.asfortranarray(xr3), np.asfortranarray(ur3), prob.fdyn, ROCKET_CONES, False)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name == f"mfmat<6,3,{N}>"
    parity_every_instance(bs.get_solution(), bs.get_status(), ref, mk, x0, kw, prob.rho, tag=f"per-instance refs N={N}")
    # (the references matter: against instance 0's for everybody the solutions differ visibly)
    assert nrel(bs.get_solution()["controls"][:, :, 1:], np.repeat(ref["u"][:, :, :1], B - 1, axis=2)) > 1e-3
    bs.close()

    # --- the workspace kept over two solves, fixed iterations: persistent oracles beside it ---
    kw = SETTINGS["fixed60"]
    mk = oracle(kw)
    orcs = [mk(b) for b in range(B)]
    bs = _solver(prob, B, kw, np.asfortranarray(xr3), np.asfortranarray(ur3), prob.fdyn, ROCKET_CONES, True)
    xs = x0.copy()
    for solve in range(2):
        bs.set_x0(xs)
        bs.solve()
        assert bs.kernel_name == f"mfmat<6,3,{N}>"
        sol = bs.get_solution()
        X, U = np.zeros_like(sol["states"]), np.zeros_like(sol["controls"])
        for b in range(B):
            orcs[b].set_x0(xs[:, b])
            orcs[b].solve()
            r = orcs[b].get_solution()
            X[:, :, b], U[:, :, b] = r["x"], r["u"]
        assert nrel_batch(sol["states"], X).max() <= FP32_TOL and nrel_batch(sol["controls"], U).max() <= FP32_TOL, \
            (solve, nrel_batch(sol["states"], X).max(), nrel_batch(sol["controls"], U).max())
        xs = np.asfortranarray(prob.A @ xs + prob.B @ U[:, 0, :] + prob.fdyn[:, None])
    for o in orcs:
        o.close()
    bs.close()


def test_mfmat_per_instance_references_chunked_and_closed_loop(hip_lib):
    """per-instance references through the other calling patterns: a chunked solve with compaction (the launch's slots are
    an index list into the batch: the tile stages the references of the instances it was handed) agrees with the
    single-launch solve, and the fused closed loop (references constant over the steps) equals the host-stepped one"""
    N, B = 10, 53
    rng = np.random.default_rng(5)
    prob = t.problems.rocket(N)
    xr, ur = t.problems.rocket_refs(N)
    xr3 = np.asfortranarray(xr[:, :, None] * (1.0 + 0.15 * rng.standard_normal((1, 1, B))))
    ur3 = np.asfortranarray(ur[:, :, None] + 0.5 * rng.standard_normal((3, N - 1, B)))
    x0 = t.problems.rocket_x0(B, seed=6)
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=5)
    outs = []
    for chunk in (0, 10):
        bs = _solver(prob, B, kw, xr3, ur3, prob.fdyn, ROCKET_CONES, True)
        if chunk:
            bs.set_compaction(chunk)
        bs.set_x0(x0)
        bs.solve()
        assert bs.kernel_name == f"mfmat<6,3,{N}>"
        outs.append((bs.get_solution(), bs.get_status()))
        bs.close()
    # (as for shared references: a chunk boundary takes the feed-forward term through the workspace's fp32 d = Quu_inv t and
    # back, so the solutions agree to the last digits, not bit for bit)
    same = outs[0][1]["iter"] == outs[1][1]["iter"]
    assert same.mean() >= 0.98
    assert nrel_batch(outs[0][0]["controls"], outs[1][0]["controls"])[same].max() <= 2e-6
    assert nrel_batch(outs[0][0]["states"], outs[1][0]["states"])[same].max() <= 2e-6
    assert len(np.unique(outs[0][1]["iter"])) > 1                      # (instances do leave at different checks: the compaction had work)
    # closed loop: fused against host-stepped
    steps = 6
    kw = dict(abs_pri_tol=2e-3, abs_dua_tol=1e-3, max_iter=40, check_termination=1)
    bs = _solver(prob, B, kw, xr3, ur3, prob.fdyn, ROCKET_CONES, True)
    bs.set_x0(x0)
    log = bs.mpc_rollout(steps)
    assert bs.kernel_name == f"mfmat<6,3,{N}>"
    bs.close()
    bs = _solver(prob, B, kw, xr3, ur3, prob.fdyn, ROCKET_CONES, True)
    xs = x0.copy()
    for k in range(steps):
        bs.set_x0(xs)
        bs.solve()
        u0 = bs.get_solution()["controls"][:, 0, :]
        assert nrel(log["u"][:, k, :], u0) <= FP32_TOL, (k, nrel(log["u"][:, k, :], u0))
        xs = np.asfortranarray(prob.A @ xs + prob.B @ u0 + prob.fdyn[:, None])
    bs.close()
