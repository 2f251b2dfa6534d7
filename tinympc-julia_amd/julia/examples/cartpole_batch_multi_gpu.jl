# Cart-pole, 2^20 problem instances over all GPUs of a node — the batched form of the reference's
# examples/cartpole_example_one_solve.jl on libtinympc_hip.so (NOT executed in the build image: it has no Julia).
#
#   julia cartpole_batch_multi_gpu.jl [n_gpus]
#
# The system and the weights are the ones the reference example states; only `batch`, `set_gpus` and the matrix-valued
# `set_x0` are new (INTEGRATION.md sections 2 and 4).
include("../TinyMPC.jl")
using .TinyMPC
using LinearAlgebra, Random

A = [1.0  0.01  0.0   0.0;
     0.0  1.0   0.039 0.0;
     0.0  0.0   1.002 0.01;
     0.0  0.0   0.458 1.002]
B = reshape([0.0; 0.02; 0.0; 0.067], 4, 1)
Q = diagm([10.0, 1.0, 10.0, 1.0])
R = diagm([1.0])
N, batch = 20, 2^20
n_gpus = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 1

prob = TinyMPCSolver()
setup(prob, A, B, zeros(4), Q, R, 1.0, 4, 1, N; batch = batch, max_iter = 100, abs_pri_tol = 1e-3, abs_dua_tol = 1e-3)
n_gpus > 1 && set_gpus(prob, n_gpus)            # contiguous shards, one per device; the solve status is folded over RCCL
set_warm_start(prob, false)                     # every solve from the zero workspace (drop this line for the reference's warm starts)
set_bound_constraints(prob, fill(-1e17, 4, N), fill(1e17, 4, N), fill(-0.5, 1, N - 1), fill(0.5, 1, N - 1))

Random.seed!(0)
X0 = vcat(0.5 .* randn(1, batch), zeros(1, batch), 0.1 .* randn(1, batch), zeros(1, batch))   # 4 x batch
set_x0(prob, X0)

t = @elapsed status = solve(prob)               # 0 iff every instance on every GPU converged
sol = get_solution(prob)                        # states 4 x N x batch, controls 1 x (N-1) x batch
st = get_status(prob)                           # per-instance iterations / solved flags / residuals
println("status $status on $(get_gpus()) GPU(s), kernel $(kernel_name()), ",
        "$(round(batch / t / 1e6, digits = 1)) M solves/s incl. host transfers; ",
        "mean iterations $(sum(st.iter) / batch), first control $(sol.controls[1, 1, 1])")
