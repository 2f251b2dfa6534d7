# The reference's rocket-landing loop (examples/rocket_landing_constraints.jl) for a BATCH of rockets, as one launch per
# 90 MPC steps: affine dynamics term, box bounds, one thrust cone and one glide-slope cone, references shifted by one knot
# per step, warm-started solves with the example's tolerances.  Problem data: rocket_landing_constraints.jl:17-57.
# (Julia is not installed in the build image: this script is written against julia/TinyMPC.jl and has not been executed;
# the same calls run in tests/test_mfmat_gpu.py::test_mfmat_fused_rocket_loop_vs_oracle through the Python mirror.)
include(joinpath(@__DIR__, "..", "TinyMPC.jl"))
using .TinyMPC
using LinearAlgebra, Random

const NSTATES, NINPUTS, NHORIZON, NTOTAL = 6, 3, 10, 100
A = [1.0 0 0 0.05 0 0; 0 1.0 0 0 0.05 0; 0 0 1.0 0 0 0.05; 0 0 0 1.0 0 0; 0 0 0 0 1.0 0; 0 0 0 0 0 1.0]
B = [0.000125 0 0; 0 0.000125 0; 0 0 0.000125; 0.005 0 0; 0 0.005 0; 0 0 0.005]
fdyn = [0.0, 0.0, -0.0122625, 0.0, 0.0, -0.4905]
Q = diagm(fill(101.0, 6)); R = diagm(fill(2.0, 3))
x_min = repeat([-5.0, -5.0, -0.5, -10.0, -10.0, -20.0], 1, NHORIZON); x_max = repeat([5.0, 5.0, 100.0, 10.0, 10.0, 20.0], 1, NHORIZON)
u_min = fill(-10.0, NINPUTS, NHORIZON - 1); u_max = fill(105.0, NINPUTS, NHORIZON - 1)

batch = 32768
solver = TinyMPCSolver()
setup(solver, A, B, fdyn, Q, R, 1.0, NSTATES, NINPUTS, NHORIZON; batch=batch, max_iter=100, abs_pri_tol=2e-3, abs_dua_tol=1e-3)
set_bound_constraints(solver, x_min, x_max, u_min, u_max)
set_cone_constraints(solver, Int32[0], Int32[3], [0.25], Int32[0], Int32[3], [0.5])     # inputs first; the workspace persists (default)

xinit, xgoal = [4.0, 2.0, 20.0, -3.0, 2.0, -4.5], zeros(6)
steps = NTOTAL - NHORIZON
x_ref_seq = zeros(NSTATES, NHORIZON, steps); u_ref_seq = zeros(NINPUTS, NHORIZON - 1, steps)
for k in 1:steps, i in 1:NHORIZON
    x_ref_seq[:, i, k] = xinit + (xgoal - xinit) * (i + k - 2) / (NTOTAL - 1)            # rocket_landing_constraints.jl:107-115
    i <= NHORIZON - 1 && (u_ref_seq[3, i, k] = 10.0)
end
set_ref_sequence(solver, x_ref_seq, u_ref_seq)

Random.seed!(2)
x0 = Float32.(1.1 .* xinit .* (1 .+ 0.05 .* (2 .* rand(NSTATES, batch) .- 1)))          # Float32 host array: a plain copy to the device
set_x0(solver, x0)
log = mpc_rollout(solver, steps)                                                         # 90 steps x 32 768 rockets, one launch
viol = count(b -> any(k -> norm(log.u[1:2, k, b]) > 0.25 * abs(log.u[3, k, b]) + 1e-4, 1:steps), 1:batch)
println("final altitude (mean): ", sum(log.x[3, end, :]) / batch, "   rockets with a thrust-cone violation: ", viol,
        "   mean ADMM iterations per step: ", sum(abs.(log.iter)) / length(log.iter))
