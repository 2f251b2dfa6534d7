"""Specialisation at setup (csrc/jit.cpp): the on-chip kernel of a shape the library was not built with is compiled once
(hipcc, a child process, from the library's own csrc/ headers), cached and loaded — the reference accepts any (nx, nu, N)
at run time (tiny_api.cpp:21-71) and should not fall to the HBM-streaming kernel for it.
CPU part: the unit compiles, links against the library, loads, and exports an entry (no GPU needed: hipcc cross-compiles).
GPU part: quadrotor N = 12 and an (8, 2, 25) family (matrix-core kernel), a (5, 2, 18) family (four lanes per instance) against the fp64 oracle,
with the calling patterns their built-in neighbours are tested with."""
import os

import numpy as np
import pytest

import tinympc_julia_amd as t
from tests.util import FP32_TOL, nrel_batch, parity_every_instance


@pytest.fixture
def jit_on(monkeypatch, tmp_path_factory):
    monkeypatch.delenv("TINYMPC_HIP_NO_JIT", raising=False)
    # one cache for the whole test session (a unit is compiled once), outside the home directory
    cache = os.environ.get("TINYMPC_TEST_JIT_CACHE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "jit_cache")
    os.makedirs(cache, exist_ok=True)
    monkeypatch.setenv("TINYMPC_HIP_CACHE", os.path.abspath(cache))
    return os.path.abspath(cache)


def test_specialise_compiles_links_and_loads_a_unit(hip_lib, jit_on):
    """(3, 1, 6): no built-in kernel -> quad<3,1,6,g4> is compiled (tens of seconds, once), loaded, and found again"""
    import glob
    assert t.specialise(4, 1, 20) is True                      # built in: nothing to do
    assert t.specialise(3, 1, 6, verbose=True) is True
    units = glob.glob(os.path.join(jit_on, "*", "quad_3_1_6_g4.so"))
    assert len(units) >= 1 and os.path.getsize(units[-1]) > 10000   # (one directory per hash of the kernel headers)
    assert t.specialise(3, 1, 6) is True                       # from the process's table now
    assert t.specialise(20, 6, 10) is False                    # beyond what a lane group / tile holds: run-time-shape kernels


def test_no_jit_switch_and_missing_compiler(hip_lib, monkeypatch, tmp_path):
    monkeypatch.setenv("TINYMPC_HIP_CACHE", str(tmp_path))
    monkeypatch.setenv("TINYMPC_HIP_NO_JIT", "1")
    assert t.specialise(5, 2, 7) is False
    monkeypatch.delenv("TINYMPC_HIP_NO_JIT")
    monkeypatch.setenv("TINYMPC_HIP_HIPCC", "/nonexistent/hipcc")
    assert t.specialise(5, 2, 7) is False                      # no compiler: fall back, no error


def _random_family(nx, nu, N, seed):
    rng = np.random.default_rng(seed)
    A = np.eye(nx) + 0.2 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= 0.97 / np.abs(np.linalg.eigvals(A)).max()
    prob = t.problems.Problem("rand", A, 0.5 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(0.5, 5.0, nx)),
                              np.diag(rng.uniform(0.5, 3.0, nu)), float(rng.uniform(0.5, 2.0)), N)
    prob.x_min, prob.x_max = np.full((nx, N), -1e17), np.full((nx, N), 1e17)
    prob.u_min, prob.u_max = np.full((nu, N - 1), -0.4), np.full((nu, N - 1), 0.4)
    return prob, rng


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["quadrotor_N12", "family_8_2_25", "family_5_2_18"])
def test_specialised_kernels_vs_oracle(hip_lib, oracle_built, jit_on, shape):
    B = 300
    if shape == "quadrotor_N12":
        prob, x0, want = t.problems.quadrotor(12), t.problems.quadrotor_x0(B, seed=6), "mfma<12,4,12>"
        xr = ur = None
    elif shape == "family_8_2_25":
        prob, rng = _random_family(8, 2, 25, 11)
        x0, want = np.asfortranarray(rng.uniform(-0.5, 0.5, (8, B))), "mfma<8,2,25>"
        xr, ur = 0.1 * rng.standard_normal((8, 25)), 0.05 * rng.standard_normal((2, 24))
    else:
        prob, rng = _random_family(5, 2, 18, 12)
        x0, want = np.asfortranarray(rng.uniform(-0.5, 0.5, (5, B))), "quad<5,2,18,g4>"
        xr, ur = 0.1 * rng.standard_normal((5, 18)), 0.05 * rng.standard_normal((2, 17))

    def oracle(kw, xb=None):
        def make(b=None):
            o = oracle_built.CpuSolver("orc64", prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N)
            o.update_settings(**kw)
            o.set_bound_constraints(prob.x_min, xb if xb is not None else prob.x_max, prob.u_min, prob.u_max)
            if xr is not None:
                o.set_x_ref(xr)
                o.set_u_ref(ur)
            return o
        return make
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    assert bs.kernel_name == want, "the shape was not specialised at setup"
    if xr is not None:
        bs.set_x_ref(xr)
        bs.set_u_ref(ur)
    for kw, warm in ((dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=1), False),
                     (dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1), False),
                     (dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=5), True)):
        bs.update_settings(**kw)
        bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        bs.set_warm_start(warm)
        bs.reset()
        bs.set_x0(x0)
        bs.solve()
        assert bs.kernel_name == want
        sol, st = bs.get_solution(), bs.get_status()
        mk = oracle(kw)
        ref = dict(x=np.zeros_like(sol["states"]), u=np.zeros_like(sol["controls"]), iter=np.zeros(B, dtype=int), solved=np.zeros(B, dtype=int), res=np.zeros((B, 4)))
        for b in range(B):
            o = mk()
            o.set_x0(x0[:, b])
            o.solve()
            r = o.get_solution()
            ref["x"][:, :, b], ref["u"][:, :, b], ref["iter"][b], ref["solved"][b], ref["res"][b] = r["x"], r["u"], r["iter"], r["solved"], r["res"]
            o.close()
        parity_every_instance(sol, st, ref, mk, x0, kw, prob.rho, min_same=0.95, tag=f"{shape} {kw}")
    # what a specialised unit does not carry goes elsewhere: adaptive rho, fp32 recurrences (4 lanes per instance)
    bs.set_adaptive_rho(True)
    bs.solve()
    assert bs.kernel_name != want
    bs.set_adaptive_rho(False)
    if shape == "family_5_2_18":
        bs.set_precision(1)
        bs.solve()
        assert bs.kernel_name.startswith("stream4<") or bs.kernel_name == "generic"
        bs.set_precision(0)
    bs.solve()
    assert bs.kernel_name == want
    bs.close()


LAYOUT_UNITS = {
    # name: (nx, nu, N, refs, cxa, cxq, cua, cuq, bv, cxa2, cxq2, cua2, cuq2, mlx, mlu)
    "two_state_cones_rows_both_sides": (6, 3, 17, 1, 0, 3, 0, 3, "false", 3, 3, 0, 0, 1, 2),
    "two_input_cones_knot_bounds": (6, 4, 17, 1, 1, 4, 0, 2, "true", 0, 0, 2, 2, 0, 0),
    "rows_only_per_instance_refs": (6, 3, 12, 2, 0, 0, 0, 0, "false", 0, 0, 0, 0, 2, 0),
    "five_states_two_inputs": (5, 2, 9, 0, 0, 3, 0, 2, "false", 3, 2, 0, 0, 0, 1),
    "eight_states": (8, 2, 12, 1, 0, 4, 0, 2, "false", 4, 4, 0, 0, 1, 0),
}


@pytest.mark.parametrize("unit", list(LAYOUT_UNITS))
def test_constraint_layout_units_compile(unit, tmp_path):
    """what csrc/jit.cpp::jit_trans_for writes for a solver whose constraint layout no built-in `mfmat` entry has — two cones on
    a side, linear rows, other shapes — compiles for gfx950 with the flags it uses (device code only: no GPU, no link), and
    the kernel holds its hand-over stores in the asm form no compiler can reorder (ds_write_b32 triples of one statement)"""
    import subprocess
    p = LAYOUT_UNITS[unit]
    csrc = os.path.join(os.path.dirname(os.path.abspath(t.__file__)), "csrc")
    src = tmp_path / "unit.hip"
    src.write_text('#include "mfmat_entry.hip.h"\nTMPC_DEFINE_MFMAT_JIT_ENTRY("mfmat<test>", ' + ", ".join(str(v) for v in p) + ")\n")
    out = tmp_path / "unit.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-honor-nans", "-DTMPC_JIT_UNIT",
                    "-DTMPC_MFMAT_HANDOVER=2", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I" + csrc, "-S", "--cuda-device-only", str(src), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    asm = out.read_text()
    assert "admm_mfmat_kernel" in asm and "v_mfma_f64_16x16x4" in asm
    import re
    # every hand-over of the rollout (N - 1 steps, every iteration's sweep is one unrolled sequence) is ONE asm statement of
    # two or three LDS stores: between #ASMSTART / #ASMEND the compiler moves nothing
    blocks = re.findall(r"#ASMSTART\n(.*?)#ASMEND", asm, flags=re.S)
    stores = [b.count("ds_write_b32") for b in blocks if "ds_write_b32" in b]
    if p[0] > 4 and p[1] >= 2 and p[0] + p[1] >= 8:           # (the mask-free stores: narrower shapes store under lane masks, owners only)
        assert len(stores) >= p[2] - 1 and set(stores) <= {2, 3}, (len(stores), set(stores))
    else:
        assert not stores
    spills = [int(m) for m in re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)]
    assert spills and max(spills) <= 160, spills             # (the built-in N = 20 kernels spill 110 under their two-waves cap)


LEAN_UNITS = {
    # name: (nx, nu, N, LIVE, UBK, ONE, XB, REFS, state type)
    "cartpole_N12_two_wavefronts": (4, 1, 12, "false", "true", "false", "false", "tmpc::REF_ZERO", "float"),
    "cartpole_N30_live_state_bound_refs": (4, 1, 30, "true", "true", "true", "true", "tmpc::REF_SHARED", "float"),
    "three_states_two_inputs": (3, 2, 16, "false", "false", "true", "false", "tmpc::REF_SHARED", "float"),
    "cartpole_N20_fp64_state_live": (4, 1, 20, "true", "true", "true", "false", "tmpc::REF_ZERO", "double"),
    "cartpole_N15_fp64_state_bound_refs": (4, 1, 15, "true", "true", "true", "true", "tmpc::REF_SHARED", "double"),
}


@pytest.mark.parametrize("unit", list(LEAN_UNITS))
def test_lean_variant_units_compile(unit, tmp_path):
    """what csrc/jit.cpp::jit_lean_for writes — ONE variant of the headline kernel for a shape without a built-in lean
    instantiation — compiles for gfx950 with its flags (device code only) and stays (nearly) spill-free"""
    import re
    import subprocess
    p = LEAN_UNITS[unit]
    csrc = os.path.join(os.path.dirname(os.path.abspath(t.__file__)), "csrc")
    src = tmp_path / "unit.hip"
    src.write_text('#include "lean_entry.hip.h"\nTMPC_DEFINE_LEAN_JIT_ENTRY("lean<test>", ' + ", ".join(str(v) for v in p) + ")\n")
    out = tmp_path / "unit.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-honor-nans", "-DTMPC_JIT_UNIT",
                    "-fno-slp-vectorize", "-I" + csrc, "-S", "--cuda-device-only", str(src), "-o", str(out)], check=True, capture_output=True, timeout=600)
    asm = out.read_text()
    assert "admm_lean_kernel" in asm
    spills = [int(m) for m in re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)]
    assert spills and max(spills) <= (48 if p[-1] == "float" else 160), spills   # (fp64 state with a state bound: 490 values per lane)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["cartpole_N12", "cartpole_N30", "family_3_2_16", "family_2_1_8"])
def test_lean_variants_specialised_at_the_first_solve(hip_lib, oracle_built, jit_on, shape):
    """one-shot solves of a cartpole-class shape WITHOUT a built-in lean instantiation run on the headline kernel all the same:
    the variant a launch needs is compiled at that launch (seconds, cached).  Every instance against the fp64 oracle in the
    calling patterns the built-in lean kernels are tested with: fixed iterations, tolerance-terminated, a finite state bound,
    shared references, input bounds that depend on the knot; other patterns (warm start) stay on the shape's quad kernel."""
    B = 20480                                                   # (from here on one lane per instance is the batch's variant)
    if shape.startswith("cartpole"):
        N = int(shape.split("N")[1])
        prob, x0 = t.problems.cartpole(N, u_bound=0.5), t.problems.cartpole_x0(B, seed=3)
        nx, nu = 4, 1
    else:
        nx, nu, N = (3, 2, 16) if shape == "family_3_2_16" else (2, 1, 8)
        prob, rng = _random_family(nx, nu, N, 14 + nx)
        x0 = np.asfortranarray(rng.uniform(-0.5, 0.5, (nx, B)))
    rng = np.random.default_rng(5)
    xr, ur = 0.1 * rng.standard_normal((nx, N)), 0.05 * rng.standard_normal((nu, N - 1))
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    cases = [("fixed", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=10), False, False, False),
             ("tol", dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1), False, False, False),
             ("state bound + refs", dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=80, check_termination=5), True, True, False),
             ("knot bounds + refs", dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=50, check_termination=0), False, True, True)]
    for tag, kw, xb, refs, knot in cases:
        x_min, x_max = prob.x_min.copy(), prob.x_max.copy()
        u_min, u_max = prob.u_min.copy(), prob.u_max.copy()
        if xb:
            x_max[0, :] = np.abs(x0[0]).max() * 0.6             # binds for the instances that start beyond it
            x_min[0, :] = -np.abs(x0[0]).max() * 0.6
        if knot:
            u_max[:, ::2] *= 0.8
        bs.update_settings(**kw)
        bs.set_bound_constraints(x_min, x_max, u_min, u_max)
        bs.set_warm_start(False)
        if refs:
            bs.set_x_ref(xr)
            bs.set_u_ref(ur)
        bs.set_x0(x0)
        bs.solve()
        assert bs.last_launch_name == f"lean<{nx},{nu},{N}>", (tag, bs.last_launch_name, bs.kernel_name)
        sol, st = bs.get_solution(), bs.get_status()
        import copy
        pb = copy.copy(prob)
        pb.x_min, pb.x_max, pb.u_min, pb.u_max = x_min, x_max, u_min, u_max
        ref = oracle_built.solve_batch("orc64", pb, x0, xref=xr if refs else None, uref=ur if refs else None, nthreads=16, **kw)

        def mk(b=None, pb=pb, kw=kw, refs=refs):
            o = oracle_built.CpuSolver("orc64", pb.A, pb.B, pb.Q, pb.R, pb.rho, pb.N)
            o.update_settings(**kw)
            o.set_bound_constraints(pb.x_min, pb.x_max, pb.u_min, pb.u_max)
            if refs:
                o.set_x_ref(xr)
                o.set_u_ref(ur)
            return o
        parity_every_instance(sol, st, ref, mk, x0, kw, prob.rho, min_same=0.95, tag=f"{shape} {tag}")
    bs.set_warm_start(True)                                     # the workspace is kept: not the lean kernel's pattern
    bs.solve()
    assert not bs.last_launch_name.startswith("lean<")
    bs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fixed", "tol", "state_bound_refs", "knot_bounds"])
def test_precision_2_runs_on_the_lean_kernel(hip_lib, oracle_built, jit_on, case):
    """tinympc_set_precision(s, 2) — the reference's own fp64 arithmetic end to end (types.hpp:15) — on the headline shape: the
    route is the generic kernel's fp64-state form (50 ms per 65 536 solves), but one-shot solves of a shape the lean kernel
    holds are launched on ITS fp64-state variant, specialised on request (0.3 ms).  x0, bounds and references are the library's
    fp32 arrays on both sides: every instance at 1e-6, iteration counts and solved flags exactly; a kept workspace stays on
    the generic kernel."""
    N, B = (15 if case == "state_bound_refs" else 20), 20480 + 37   # (fp64 slack AND dual of every state row: the registers hold N <= 17)
    f32 = lambda a: np.asfortranarray(np.asarray(a, dtype=np.float32).astype(np.float64))
    prob, x0 = t.problems.cartpole(N, u_bound=0.5), f32(t.problems.cartpole_x0(B, seed=9))
    rng = np.random.default_rng(3)
    xr = ur = None
    kw = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=10)
    if case != "fixed":
        kw = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
    if case == "state_bound_refs":
        prob.x_min, prob.x_max = prob.x_min.copy(), prob.x_max.copy()
        prob.x_min[0, :], prob.x_max[0, :] = -0.25, 0.25
        xr, ur = f32(0.1 * rng.standard_normal((4, N))), f32(0.05 * rng.standard_normal((1, N - 1)))
    if case == "knot_bounds":
        prob.u_max = prob.u_max.copy()
        prob.u_max[:, ::2] = 0.375
    ref = oracle_built.solve_batch("orc64", prob, x0, xref=xr, uref=ur, nthreads=16, **kw)
    bs = t.BatchSolver(prob.A, prob.B, prob.Q, prob.R, prob.rho, prob.N, batch=B)
    bs.update_settings(**kw)
    bs.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if xr is not None:
        bs.set_x_ref(xr)
        bs.set_u_ref(ur)
    bs.set_warm_start(False)
    bs.set_precision(2)
    bs.set_x0(x0)
    bs.solve()
    assert bs.kernel_name == "generic<f64>" and bs.last_launch_name == f"lean<4,1,{N};f64>", (bs.kernel_name, bs.last_launch_name)
    sol, st = bs.get_solution(), bs.get_status()
    assert np.array_equal(st["iter"], ref["iter"]) and np.array_equal(st["solved"], ref["solved"])
    assert nrel_batch(sol["states"], ref["x"]).max() <= 1e-6 and nrel_batch(sol["controls"], ref["u"]).max() <= 1e-6
    assert np.abs(st["residuals"] - ref["res"]).max() <= 1e-6 * max(1.0, np.abs(ref["res"]).max())
    bs.set_warm_start(True)                                     # the workspace is kept: the generic kernel's pattern
    bs.solve()
    assert bs.last_launch_name == "generic<f64>"
    bs.set_precision(0)                                         # and back: the fp32-state lean kernel of the library
    bs.set_warm_start(False)
    bs.solve()
    assert bs.last_launch_name == f"lean<4,1,{N}>"
    bs.close()


def test_concurrent_specialisation_of_one_unit(hip_lib, tmp_path, monkeypatch):
    """one process per GPU is the deployment model: four processes ask for the same new unit at the same moment (a fresh
    cache) — every one of them gets it, and what is left behind is one object and no half-written files"""
    import glob
    import subprocess
    import sys
    monkeypatch.delenv("TINYMPC_HIP_NO_JIT", raising=False)
    env = dict(os.environ, TINYMPC_HIP_CACHE=str(tmp_path))
    env.pop("TINYMPC_HIP_NO_JIT", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"import sys; sys.path.insert(0, {root!r}); import tinympc_julia_amd as t; print('RESULT', t.specialise(3, 1, 7))"
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(4)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all("RESULT True" in o for o, _ in outs), outs
    assert len(glob.glob(os.path.join(str(tmp_path), "*", "quad_3_1_7_g4.so"))) == 1
    assert not glob.glob(os.path.join(str(tmp_path), "*", "*.tmp.*"))
