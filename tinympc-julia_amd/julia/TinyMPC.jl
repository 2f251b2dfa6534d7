module TinyMPC

# Julia host side of the MI355X batched TinyMPC engine.
#
# Keeps the surface of the reference module (reference: src/TinyMPC.jl:3-6,34-292):
#   TinyMPCSolver / setup / set_x0 / set_x_ref / set_u_ref / set_bound_constraints /
#   update_settings / set_cache_terms / solve / get_solution
# and reaches the solver the same way — `ccall((:sym, libpath), Int32, ...)` by symbol name into a
# C-ABI shared library — but the library is libtinympc_hip.so (include/tinympc_hip.h) instead of
# libtinympc_jl built from src/bindings.cpp.  A batch dimension is added:
#   setup(...; batch=B)
#   set_x0(solver, x0)        x0 :: Vector (broadcast) or Matrix (nx, B)
#   set_x_ref(solver, xref)   Matrix (nx, N) shared, or Array{Float64,3} (nx, N, B)
#   get_solution(solver)      states (nx, N) for B == 1 (the reference's shape), else (nx, N, B)
#   get_status(solver)        per-instance iter / solved / residuals
#
# NOTE: the Julia toolchain is absent from the image this repository is built and tested in, so this
# file has not been executed there; the C-ABI it binds is exercised call-for-call by the Python
# mirror (tinympc-julia_amd/tinympc.py) in tests/.  INTEGRATION.md shows the two-line change that
# makes the reference's own src/TinyMPC.jl use this library without adopting this module.

export TinyMPCSolver, setup, solve, get_solution, get_solution!, pin_host!, unpin_host!, get_status, set_x0, set_x_ref, set_u_ref, set_ref_sequence, mpc_rollout,
       set_bound_constraints, set_linear_constraints, set_equality_constraints, set_cone_constraints, update_settings,
       set_cache_terms, set_batch_size, set_gpus, get_gpus, set_warm_start, set_precision, kernel_name, reset_workspace, print_problem_data,
       compute_sensitivity_autograd, set_sensitivity, get_adaptive_rho

using LinearAlgebra, Libdl, Printf

const _lib = Ref{String}(joinpath(dirname(@__DIR__), "lib", "libtinympc_hip." * Libdl.dlext))
_lib_path() = _lib[]
set_library_path!(p::AbstractString) = (_lib[] = String(p))
_ensure_loaded() = isfile(_lib_path()) || error("TinyMPC library not found: $(_lib_path())")
_last_error() = unsafe_string(ccall((:tinympc_last_error, _lib_path()), Cstring, ()))

mutable struct TinyMPCSolver
    nx::Int
    nu::Int
    N::Int
    batch::Int
    rho::Float64
    is_setup::Bool
    A::Matrix{Float64}
    B::Matrix{Float64}
    Q::Matrix{Float64}
    R::Matrix{Float64}
    TinyMPCSolver() = new(0, 0, 0, 1, 0.0, false, zeros(0, 0), zeros(0, 0), zeros(0, 0), zeros(0, 0))
end

# ---- small helpers ---------------------------------------------------------------------------
# every matrix crosses the boundary as (pointer, rows, cols), column-major Float64 — Julia's own layout
_mat(a::AbstractMatrix{Float64}) = Matrix(a)
_mat(a::AbstractVector{Float64}) = reshape(collect(a), length(a), 1)
_mat(a::AbstractArray{Float64,3}) = reshape(Array(a), size(a, 1), size(a, 2) * size(a, 3))  # (rows, knots*batch)
_flag(b::Bool) = Int32(b ? 1 : 0)
_need(solver) = solver.is_setup || error("Solver not setup")
_ok(status, what) = status == 0 ? status : error("$what ($(_last_error()))")

"""
    setup(solver, A, B, f, Q, R, rho, nx, nu, N; batch=1, kwargs...)

Same call as the reference (src/TinyMPC.jl:55-112) plus `batch`.  The infinite-horizon Riccati
precompute runs on the host in Float64 inside the library; a non-zero `f` (affine dynamics) is honoured
but its parity with the upstream solver is unpinned (DESIGN.md §6).  Settings are then
pushed with every `en_*` flag false, exactly like the reference (src/TinyMPC.jl:89-104).
"""
function setup(solver::TinyMPCSolver, A::Matrix{Float64}, B::Matrix{Float64}, f::Vector{Float64},
               Q::Matrix{Float64}, R::Matrix{Float64}, rho::Float64, nx::Int, nu::Int, N::Int;
               batch::Int=1, verbose::Bool=false, abs_pri_tol::Float64=1e-3, abs_dua_tol::Float64=1e-3,
               max_iter::Int=100, check_termination::Bool=true, adaptive_rho::Bool=false,
               adaptive_rho_min::Float64=0.1, adaptive_rho_max::Float64=10.0,
               adaptive_rho_clipping::Bool=true)
    _ensure_loaded()
    solver.nx, solver.nu, solver.N, solver.rho, solver.batch = nx, nu, N, rho, batch
    solver.A, solver.B, solver.Q, solver.R = copy(A), copy(B), copy(Q), copy(R)
    fm = _mat(f)
    status = ccall((:setup_solver, _lib_path()), Int32,
                   (Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32,
                    Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32, Float64, Int32, Int32, Int32, Int32),
                   A, size(A, 1), size(A, 2), B, size(B, 1), size(B, 2), fm, size(fm, 1), size(fm, 2),
                   Q, size(Q, 1), size(Q, 2), R, size(R, 1), size(R, 2), rho, nx, nu, N, _flag(verbose))
    status == 0 || error("Setup failed with status: $status ($(_last_error()))")
    batch == 1 || _ok(ccall((:set_batch_size, _lib_path()), Int32, (Int32,), batch), "Failed to set batch size")
    solver.is_setup = true
    update_settings(solver; abs_pri_tol=abs_pri_tol, abs_dua_tol=abs_dua_tol, max_iter=max_iter,
                    check_termination=check_termination, adaptive_rho=adaptive_rho,
                    adaptive_rho_min=adaptive_rho_min, adaptive_rho_max=adaptive_rho_max,
                    adaptive_rho_enable_clipping=adaptive_rho_clipping, verbose=verbose)
    verbose && @printf("TinyMPC solver setup successful (nx=%d, nu=%d, N=%d, batch=%d)\n", nx, nu, N, batch)
    return status
end

function set_batch_size(solver::TinyMPCSolver, batch::Int)
    _need(solver)
    _ok(ccall((:set_batch_size, _lib_path()), Int32, (Int32,), batch), "Failed to set batch size")
    solver.batch = batch
    return 0
end

# Spread the solver's batch over GPUs 0 .. n-1 of the node in contiguous shards (n = 1: back to one device).  Every
# other call is unchanged and acts on the whole batch; solve() returns 0 iff every instance on every GPU converged
# (the status block is folded over RCCL).  include/tinympc_hip.h section 3, INTEGRATION.md section 4.
function set_gpus(solver::TinyMPCSolver, n::Integer)
    _need(solver)
    _ok(ccall((:set_gpus, _lib_path()), Int32, (Int32,), Int32(n)), "Failed to set the number of GPUs")
    return 0
end
get_gpus() = Int(ccall((:get_gpus, _lib_path()), Int32, ()))

# on = false: every solve() starts from the zero workspace and keeps none (one-shot solves: the benchmark regime, served by
# the on-chip kernels); true (default): the reference's semantics, the workspace persists between solves
# Arithmetic of the solver: 0 = fp64 recurrences over fp32 state (default, 1e-5 of the reference), 1 = all fp32 (faster on
# shapes without a matrix-core kernel, does not hold 1e-5 on every instance), 2 = fp64 end to end like the reference
# (types.hpp:15) — for validation against the CPU solver; slow.
function set_precision(solver::TinyMPCSolver, precision::Integer)
    _need(solver)
    _ok(ccall((:set_precision, _lib_path()), Int32, (Int32,), precision), "Failed to set precision")
end

function set_warm_start(solver::TinyMPCSolver, on::Bool)
    _need(solver)
    _ok(ccall((:set_warm_start, _lib_path()), Int32, (Int32,), on ? 1 : 0), "Failed to set warm start")
    return 0
end
kernel_name() = unsafe_string(ccall((:get_kernel_name, _lib_path()), Cstring, ()))

# x0: Vector (length nx, broadcast to the batch) or Matrix (nx, batch)
function set_x0(solver::TinyMPCSolver, x0::AbstractVecOrMat{Float64}; verbose::Bool=false)
    _need(solver)
    m = _mat(x0)
    _ok(ccall((:set_x0, _lib_path()), Int32, (Ptr{Float64}, Int32, Int32, Int32),
              m, size(m, 1), size(m, 2), _flag(verbose)), "Failed to set initial state")
end

# Float32 host arrays: the device buffers are fp32, so this is a plain copy (the Float64 method narrows 65 536 x nx values on
# the host first).  x0: (nx,) broadcast to the batch, or (nx, batch)
function set_x0(solver::TinyMPCSolver, x0::AbstractVecOrMat{Float32}; verbose::Bool=false)
    _need(solver)
    m = x0 isa AbstractVector ? reshape(collect(x0), :, 1) : Matrix{Float32}(x0)
    _ok(ccall((:set_x0_f32, _lib_path()), Int32, (Ptr{Float32}, Int32, Int32, Int32),
              m, size(m, 1), size(m, 2), _flag(verbose)), "Failed to set initial state")
end

# x_ref: Matrix (nx, N) shared by the batch, or Array{Float64,3} (nx, N, batch)
function set_x_ref(solver::TinyMPCSolver, x_ref::AbstractArray{Float64}; verbose::Bool=false)
    _need(solver)
    m = _mat(x_ref)
    _ok(ccall((:set_x_ref, _lib_path()), Int32, (Ptr{Float64}, Int32, Int32, Int32),
              m, size(m, 1), size(m, 2), _flag(verbose)), "Failed to set state reference")
end

# u_ref: Matrix (nu, N-1) shared by the batch, or Array{Float64,3} (nu, N-1, batch)
function set_u_ref(solver::TinyMPCSolver, u_ref::AbstractArray{Float64}; verbose::Bool=false)
    _need(solver)
    m = _mat(u_ref)
    _ok(ccall((:set_u_ref, _lib_path()), Int32, (Ptr{Float64}, Int32, Int32, Int32),
              m, size(m, 1), size(m, 2), _flag(verbose)), "Failed to set input reference")
end

# Returns 0 (every instance converged), 1 (some instance hit max_iter) or -1; never throws on it,
# like the reference (src/TinyMPC.jl:143-148).
function solve(solver::TinyMPCSolver; verbose::Bool=false)
    _need(solver)
    return ccall((:solve_mpc, _lib_path()), Int32, (Int32,), _flag(verbose))
end

function get_solution(solver::TinyMPCSolver)
    _need(solver)
    nx, nu, N, B = solver.nx, solver.nu, solver.N, solver.batch
    xs, us = zeros(nx * N * B), zeros(nu * (N - 1) * B)
    xr, xc, ur, uc = Ref{Int32}(), Ref{Int32}(), Ref{Int32}(), Ref{Int32}()
    s1 = ccall((:get_states, _lib_path()), Int32, (Ptr{Float64}, Ref{Int32}, Ref{Int32}), xs, xr, xc)
    s2 = ccall((:get_controls, _lib_path()), Int32, (Ptr{Float64}, Ref{Int32}, Ref{Int32}), us, ur, uc)
    (s1 != 0 || s2 != 0) && error("Failed to get solution ($(_last_error()))")
    B == 1 && return (states=reshape(xs, nx, N), controls=reshape(us, nu, N - 1))   # the reference's shapes
    return (states=reshape(xs, nx, N, B), controls=reshape(us, nu, N - 1, B))
end

# The solution into the caller's own Float32 arrays (nx, N, batch) / (nu, N-1, batch): no allocation, no fp32 -> fp64
# widening on the host — the per-solve round trip of a closed loop that keeps Float32 state.
# (Device-resident callers skip the copies altogether: `device_buffers(solver)` returns the fp32 device pointers of x0,
# the references, states, controls and the status arrays; with AMDGPU.jl wrap them with
# `unsafe_wrap(ROCArray, convert(Ptr{Float32}, ptr), dims)`, fill x0 on the device, and call `solve_async` / `solve`.)
function get_solution!(states::Array{Float32,3}, controls::Array{Float32,3}, solver::TinyMPCSolver)
    _need(solver)
    nx, nu, N, B = solver.nx, solver.nu, solver.N, solver.batch
    (size(states) == (nx, N, B) && size(controls) == (nu, N - 1, B)) || error("get_solution!: states (nx, N, batch), controls (nu, N-1, batch)")
    xr, xc, ur, uc = Ref{Int32}(), Ref{Int32}(), Ref{Int32}(), Ref{Int32}()
    s1 = ccall((:get_states_f32, _lib_path()), Int32, (Ptr{Float32}, Ref{Int32}, Ref{Int32}), states, xr, xc)
    s2 = ccall((:get_controls_f32, _lib_path()), Int32, (Ptr{Float32}, Ref{Int32}, Ref{Int32}), controls, ur, uc)
    (s1 != 0 || s2 != 0) && error("Failed to get solution ($(_last_error()))")
    return (states=states, controls=controls)
end

# Page-lock an array that is reused with `get_solution!` / the Float32 `set_x0` (the copies then DMA straight into it).  The
# library never pins caller memory by itself: call `unpin_host!` before the array can be garbage-collected, e.g.
#   pin_host!(solver, states); try ... finally unpin_host!(solver, states) end
function pin_host!(solver::TinyMPCSolver, a::Array{Float32})
    _need(solver)
    _ok(ccall((:pin_host_buffer, _lib_path()), Int32, (Ptr{Cvoid}, Csize_t), a, sizeof(a)), "Failed to pin host array")
end
function unpin_host!(solver::TinyMPCSolver, a::Array{Float32})
    _need(solver)
    _ok(ccall((:unpin_host_buffer, _lib_path()), Int32, (Ptr{Cvoid},), a), "Failed to unpin host array")
end

# Per-instance iteration count, solved flag and residuals (pri_state, dua_state, pri_input, dua_input)
function get_status(solver::TinyMPCSolver)
    _need(solver)
    B = solver.batch
    iter, solved, res = zeros(Int32, B), zeros(Int32, B), zeros(4, B)
    _ok(ccall((:get_status, _lib_path()), Int32, (Ptr{Int32}, Ptr{Int32}, Ptr{Float64}), iter, solved, res),
        "Failed to get status")
    return (iter=iter, solved=solved, residuals=res)
end

# Shared references of every step of the next fused closed loops: x_ref_seq (nx, N, steps), u_ref_seq (nu, N-1, steps) —
# what the loop of examples/rocket_landing_constraints.jl:107-115 passes to set_x_ref / set_u_ref step by step
function set_ref_sequence(solver::TinyMPCSolver, x_ref_seq::Array{Float64,3}, u_ref_seq::Array{Float64,3})
    _need(solver)
    steps = size(x_ref_seq, 3)
    (size(x_ref_seq) == (solver.nx, solver.N, steps) && size(u_ref_seq) == (solver.nu, solver.N - 1, steps)) ||
        error("set_ref_sequence: x_ref_seq (nx, N, steps), u_ref_seq (nu, N-1, steps)")
    _ok(ccall((:set_ref_sequence, _lib_path()), Int32, (Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32, Int32),
              x_ref_seq, solver.nx, solver.N * steps, u_ref_seq, solver.nu, (solver.N - 1) * steps, steps),
        "Failed to set the reference sequence")
end

# `steps` closed-loop MPC steps in ONE launch: solve -> u0 = controls[:, 1] -> x0 = A x0 + B u0 + f -> next solve, the
# warm-start workspace staying on chip (the host loops of cartpole_example_mpc.jl:35-51, rocket_landing_constraints.jl:97-134).
# Returns (status of the last solve, x (nx, steps, batch) plant states, u (nu, steps, batch) applied controls,
# iter (steps, batch) ADMM iterations per step, negative where the step hit max_iter).
function mpc_rollout(solver::TinyMPCSolver, steps::Integer)
    _need(solver)
    nx, nu, B = solver.nx, solver.nu, solver.batch
    x, u, it = zeros(nx, steps, B), zeros(nu, steps, B), zeros(Int32, steps, B)
    st = ccall((:mpc_rollout, _lib_path()), Int32, (Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}), steps, x, u, it)
    st < 0 && error("mpc_rollout failed ($(_last_error()))")
    return (status=st, x=x, u=u, iter=it)
end

function reset_workspace(solver::TinyMPCSolver)
    _need(solver)
    _ok(ccall((:reset_workspace, _lib_path()), Int32, ()), "Failed to reset workspace")
end

# Like the reference (src/TinyMPC.jl:181-211) every field is sent with its keyword default, so a later
# call resets en_*_bound to false and the tolerances to 1e-3.
function update_settings(solver::TinyMPCSolver;
                         abs_pri_tol::Float64=1e-3, abs_dua_tol::Float64=1e-3, max_iter::Int=100,
                         check_termination::Bool=true, en_state_bound::Bool=false, en_input_bound::Bool=false,
                         en_state_soc::Bool=false, en_input_soc::Bool=false, en_state_linear::Bool=false,
                         en_input_linear::Bool=false, adaptive_rho::Bool=false, adaptive_rho_min::Float64=0.1,
                         adaptive_rho_max::Float64=10.0, adaptive_rho_enable_clipping::Bool=true,
                         verbose::Bool=false)
    _ok(ccall((:update_settings, _lib_path()), Int32,
              (Float64, Float64, Int32, Int32, Int32, Int32, Int32, Int32, Int32, Int32,
               Int32, Float64, Float64, Int32, Int32),
              abs_pri_tol, abs_dua_tol, max_iter, _flag(check_termination), _flag(en_state_bound),
              _flag(en_input_bound), _flag(en_state_soc), _flag(en_input_soc), _flag(en_state_linear),
              _flag(en_input_linear), _flag(adaptive_rho), adaptive_rho_min, adaptive_rho_max,
              _flag(adaptive_rho_enable_clipping), _flag(verbose)), "Failed to update settings")
end

# per-knot bounds (nx, N) / (nu, N-1), shared by the batch; the library enables both bound flags
function set_bound_constraints(solver::TinyMPCSolver, x_min::Matrix{Float64}, x_max::Matrix{Float64},
                               u_min::Matrix{Float64}, u_max::Matrix{Float64}; verbose::Bool=false)
    _ok(ccall((:set_bound_constraints, _lib_path()), Int32,
              (Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32,
               Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32, Int32),
              x_min, size(x_min, 1), size(x_min, 2), x_max, size(x_max, 1), size(x_max, 2),
              u_min, size(u_min, 1), size(u_min, 2), u_max, size(u_max, 1), size(u_max, 2), _flag(verbose)),
        "Failed to set bound constraints")
end

# Linear inequalities Alin_x x <= blin_x, Alin_u u <= blin_u at every knot (at most 8 rows per side); equalities as
# two opposite rows each.  Cones: per-knot second-order cones, inputs first,
# 0-based first row `Ac`, dimension `qc`, slope `c` (last row of the block is the axis); parity unpinned.
function set_linear_constraints(solver::TinyMPCSolver, Alin_x::Matrix{Float64}, blin_x::Vector{Float64},
                                Alin_u::Matrix{Float64}, blin_u::Vector{Float64}; verbose::Bool=false)
    _ok(ccall((:set_linear_constraints, _lib_path()), Int32,
              (Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32),
              Alin_x, size(Alin_x, 1), size(Alin_x, 2), blin_x, length(blin_x),
              Alin_u, size(Alin_u, 1), size(Alin_u, 2), blin_u, length(blin_u), _flag(verbose)),
        "Failed to set linear constraints")
end

function set_equality_constraints(solver::TinyMPCSolver, Aeq_x::Matrix{Float64}, beq_x::Vector{Float64};
                                  Aeq_u::Matrix{Float64}=zeros(0, solver.nu), beq_u::Vector{Float64}=zeros(Float64, 0))
    return set_linear_constraints(solver, vcat(Aeq_x, -Aeq_x), vcat(beq_x, -beq_x), vcat(Aeq_u, -Aeq_u), vcat(beq_u, -beq_u))
end

function set_cone_constraints(solver::TinyMPCSolver, Acu::Vector{Int32}, qcu::Vector{Int32}, cu::Vector{Float64},
                              Acx::Vector{Int32}, qcx::Vector{Int32}, cx::Vector{Float64}; verbose::Bool=false)
    _ok(ccall((:set_cone_constraints, _lib_path()), Int32,
              (Ptr{Int32}, Int32, Ptr{Int32}, Int32, Ptr{Float64}, Int32,
               Ptr{Int32}, Int32, Ptr{Int32}, Int32, Ptr{Float64}, Int32, Int32),
              Acu, length(Acu), qcu, length(qcu), cu, length(cu),
              Acx, length(Acx), qcx, length(qcx), cx, length(cx), _flag(verbose)),
        "Failed to set cone constraints")
end

function set_cache_terms(solver::TinyMPCSolver, Kinf::Matrix{Float64}, Pinf::Matrix{Float64},
                         Quu_inv::Matrix{Float64}, AmBKt::Matrix{Float64}; verbose::Bool=false)
    _need(solver)
    _ok(ccall((:set_cache_terms, _lib_path()), Int32,
              (Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32,
               Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32, Int32),
              Kinf, size(Kinf, 1), size(Kinf, 2), Pinf, size(Pinf, 1), size(Pinf, 2),
              Quu_inv, size(Quu_inv, 1), size(Quu_inv, 2), AmBKt, size(AmBKt, 1), size(AmBKt, 2), _flag(verbose)),
        "Failed to set cache terms")
end

# LQR gains of the rho-regularised problem the sensitivities are differenced on (reference TinyMPC.jl:326-352):
# rho enters once, P starts at Q + rho I, a 1e-8 ridge in the gain solve only, at most 5000 sweeps.
function _lqr_gains(A, B, Q, R, rho)
    nx, nu = size(A, 1), size(B, 2)
    Qr = Q + rho * Matrix{Float64}(I, nx, nx)
    Rr = R + rho * Matrix{Float64}(I, nu, nu)
    P = copy(Qr)
    K = zeros(nu, nx)
    for sweep in 1:5000
        Kold = K
        K = (Rr + B' * P * B + 1e-8 * Matrix{Float64}(I, nu, nu)) \ (B' * P * A)
        P = Qr + A' * P * (A - B * K)
        (sweep > 1 && norm(K - Kold) < 1e-10) && break
    end
    return K, P, inv(Rr + B' * P * B), Matrix((A - B * K)')
end

"""
    compute_sensitivity_autograd(solver) -> (dK, dP, dC1, dC2)

Forward differences (h = 1e-6) of the cache terms in rho, as the reference's function of the same name
(TinyMPC.jl:301-323).  `update_settings(adaptive_rho=true)` without `set_sensitivity` makes the library do the same.
"""
function compute_sensitivity_autograd(solver::TinyMPCSolver)
    _need(solver)
    h = 1e-6
    g0 = _lqr_gains(solver.A, solver.B, solver.Q, solver.R, solver.rho)
    g1 = _lqr_gains(solver.A, solver.B, solver.Q, solver.R, solver.rho + h)
    return ntuple(i -> (g1[i] - g0[i]) / h, 4)
end

# What codegen_with_sensitivity (reference TinyMPC.jl:374-394) bakes into generated code, for the live solver.
function set_sensitivity(solver::TinyMPCSolver, dK::Matrix{Float64}, dP::Matrix{Float64},
                         dC1::Matrix{Float64}, dC2::Matrix{Float64}; verbose::Bool=false)
    _need(solver)
    _ok(ccall((:set_sensitivity, _lib_path()), Int32,
              (Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32,
               Ptr{Float64}, Int32, Int32, Ptr{Float64}, Int32, Int32, Int32),
              dK, size(dK, 1), size(dK, 2), dP, size(dP, 1), size(dP, 2),
              dC1, size(dC1, 1), size(dC1, 2), dC2, size(dC2, 1), size(dC2, 2), _flag(verbose)),
        "Failed to set sensitivity matrices")
end

# rho of every instance after adaptation
function get_adaptive_rho(solver::TinyMPCSolver)
    _need(solver)
    rho = zeros(Float64, solver.batch)
    n = Ref{Int32}(0)
    _ok(ccall((:get_adaptive_rho, _lib_path()), Int32, (Ptr{Float64}, Ptr{Int32}), rho, n), "Failed to get adaptive rho")
    return rho
end

function print_problem_data(solver::TinyMPCSolver; verbose::Bool=false)
    _need(solver)
    ccall((:print_problem_data, _lib_path()), Int32, (Int32,), _flag(verbose))
end

cleanup() = try; ccall((:cleanup_solver, _lib_path()), Cvoid, ()); catch; end
atexit(cleanup)

end # module
