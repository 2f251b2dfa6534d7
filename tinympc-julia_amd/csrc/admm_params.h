// Shared host/device definitions for the fused ADMM kernels.
#pragma once
#include <stdint.h>

namespace tmpc {

// How reference trajectories reach the kernel.
enum RefMode : int {
    REF_ZERO = 0,          // Xref = Uref = 0 (what tiny_setup leaves, tiny_api.cpp:102-103)
    REF_SHARED = 1,        // one (nx,N)/(nu,N-1) pair broadcast over the batch
    REF_PER_INSTANCE = 2,  // [B][N][nx] / [B][N-1][nu]
};

// Slots of the per-solve device status block (uint32 each).
//   [0..3] max over instances of (pri_x, dua_x, pri_u, dua_u), float bits (non-negative
//          floats order like unsigned ints, so atomicMax on the bits is a float max)
//   [4]    number of instances that hit max_iter without converging
//   [5]    number of instances whose solution contains a non-finite value
enum { GSTAT_WORDS = 8 };

// generic (runtime-shape) kernel limits
constexpr int LIN_MAX_ROWS = 8, GEN_MAX_NX = 64;
constexpr int GEN_MAX_NU = 32;

struct AdmmParams {
    // family constants (device pointers)
    const float *coef;    // lane-role-major coefficient pack, layout in QuadShape / generic kernel
    const float *bounds;  // per-knot bounds pack
    // per-instance inputs
    const float *x0;    // [B][nx]
    const double *x0d;  // matrix-core kernel, closed loop: the plant state in fp64 (NULL: x0)
    const float *xref;  // REF_SHARED: [N][nx]   REF_PER_INSTANCE: [B][N][nx]
    const float *uref;  // REF_SHARED: [N-1][nu] REF_PER_INSTANCE: [B][N-1][nu]
    // per-instance outputs
    float *xout;   // [B][N][nx]    projected slack vnew (admm.cpp:187,204)
    float *uout;   // [B][N-1][nu]  projected slack znew (admm.cpp:188,205)
    int *iter;     // [B]
    int *solved;   // [B]
    float *res;    // [B][4] pri_x, dua_x, pri_u, dua_u
    // warm-start state, persists between solves: d,y,z [B][N-1][nu]; g,v [B][N][nx]
    float *sd, *sy, *sz, *sg, *sv;
    uint32_t *gstat;  // [GSTAT_WORDS] status block of the launch: max residual bits [0..3], unsolved count [4]
    uint32_t *gacc;   // [GSTAT_WORDS] accumulator behind it (fold_status); zero between launches
    // stream / generic kernels: per-instance scratch in HBM
    float *scratch;
    // chunked solves with compaction (Solver::solve_chunked): launch slot j works on instance idx[j] (NULL: j) and
    // reports iter_offset + its own iteration count; `batch` is then the number of slots of this launch
    const int *idx;
    int iter_offset;
    int batch;
    int max_iter;
    int check_termination;  // <= 0: never check (the reference divides by it, admm.cpp:91)
    int ref_mode;
    int cold_start;  // 1: start from the zero workspace, do not read sd..sv
    int save_state;  // 1: write sd..sv back at exit
    float abs_pri_tol, abs_dua_tol, rho;
    int nx, nu, N;  // run-time shape (stream kernel: N; generic kernel: all three)
    // fused closed loop (0 = plain solve): steps per launch and per-step logs
    int mpc_steps;
    float *mpc_x;    // [B][steps][nx]  plant state after each step
    float *mpc_u;    // [B][steps][nu]  control applied at each step
    int *mpc_iter;   // [B][steps]      ADMM iterations of each step, negative if it hit max_iter
    float *x0_out;   // [B][nx]         plant state after the last step (aliases x0)
    // ---- stream / generic kernels: affine dynamics + second-order cones (parity UNPINNED) ----
    int xb_active;           // some enabled state bound is finite (else vnew = x + g is never clamped)
    int has_fdyn;            // coef pack carries fdyn, APf, BPf behind the matrices
    int ncx, ncu;            // number of state / input cones per knot (0: disabled), at most 8 each
    int Acx[8], qcx[8], Acu[8], qcu[8];  // first row and dimension of each cone block
    float cx[8], cu[8];                  // mu of each cone: ||head|| <= mu * (last row)
    float *sgc, *svc, *syc, *szc;        // warm-start state of the cone slack/dual pairs
    // ---- linear inequalities Alin_x x <= blin_x, Alin_u u <= blin_u at every knot (parity UNPINNED) ----
    int mlx, mlu;                        // rows per side (0: disabled), at most LIN_MAX_ROWS each
    const float *lin;                    // [mlx][nx] rows | b[mlx] | |a|^2[mlx] | [mlu][nu] rows | b[mlu] | |a|^2[mlu]
    float *sgl, *svl, *syl, *szl;        // warm-start state of the linear-inequality slack/dual pairs
    // ---- stream / generic kernels: adaptive rho (admm.cpp:147-174, rho_benchmark.cpp) ----
    int adaptive_rho;                    // every 5th iteration each instance re-predicts its rho and Taylor-updates Kinf, Pinf
    int rho_clip;
    double rho_family;                   // the family's rho as a double (the adaptive state's reset value; `rho` above is a float)
    float rho_min, rho_max;
    const double *sens;                  // dKinf/drho [nu*nx] then dPinf/drho [nx*nx], column-major
    double *adapt;                       // [1 + nu*nx + nx*nx][batch]: rho, Kinf, Pinf of each instance; solver state, it
                                         // persists between solves like the reference's cache
    long adapt_stride;                   // instances per row of `adapt` (the solver's batch)
    void *adp_cols;                      // stream kernel, adaptive rho: [ADP_LEN][G * batch] scratch columns (kernel-local)
    // ---- stream kernel only: one problem family PER INSTANCE ----
    const float *het_aux;                // [nx + nu + 1][batch]: diag(Q)+rho, diag(R)+rho, rho of each instance
    // ---- mfmac kernel only: 0 = the bounds pack holds one knot's bounds (they do not depend on the knot) ----
    int bounds_stride;
    // ---- mfmat kernel, fused closed loop: shared references of every step, [steps][N][nx] / [steps][N-1][nu] (NULL: the
    // references stay as set) — the per-step shift of rocket_landing_constraints.jl:107-115 ----
    const float *xref_seq, *uref_seq;
    // ---- lean kernel (admm_lean.hip.h): its fp64 coefficient pack (LeanPack), wave-uniform ----
    const double *lean;
    // ---- launcher-side switches (read from the environment once per solver: Switches), not read by any kernel ----
    int host_flags;   // HF_NO_REFILL | HF_NO_UNI | HF_NO_OS | HF_LEAN_ONE
    // ---- generic kernel, precision 2 (fp64 end to end): the workspace kept between solves (Ws64) and the tolerances in fp64 ----
    double *ws64;
    double abs_pri_tol64, abs_dua_tol64;
};
enum : int { HF_NO_REFILL = 1, HF_NO_UNI = 2, HF_NO_OS = 4, HF_LEAN_ONE = 8 };

// Coefficient pack of the lean kernel (admm_lean.hip.h), fp64, all wave-uniform; filled by build_lean_pack (kernels.hip).
struct LeanLayout {
    int oM;      // A - B Kinf     [nx][nx] row-major (its transpose is the AmBKt the backward sweep reads)
    int oK;      // Kinf           [nu][nx]
    int oB;      // B              [nx][nu]
    int oC;      // -rho Quu_inv   [nu][nu]
    int len;
    int padded;  // rounded up to whole 8-double scalar loads (what the kernel keeps in SGPRs)
    int oP;      // Pinf           [nx][nx] row-major, behind the padded block (read once, for the terminal reference term)
    int total;
};
constexpr LeanLayout lean_layout(int nx, int nu) {
    return LeanLayout{0, nx * nx, nx * nx + nu * nx, nx * nx + 2 * nu * nx, nx * nx + 2 * nu * nx + nu * nu,
                      (nx * nx + 2 * nu * nx + nu * nu + 7) / 8 * 8, (nx * nx + 2 * nu * nx + nu * nu + 7) / 8 * 8,
                      (nx * nx + 2 * nu * nx + nu * nu + 7) / 8 * 8 + nx * nx};
}

#ifdef __HIPCC__
// Folds a workgroup's residual maxima / unsolved count into the launch's status block without a host-side clear: the
// wavefronts of a workgroup (at most four) meet in LDS, ONE lane per workgroup accumulates in P.gacc (per-wavefront atomics
// queue on five words when a thousand wavefronts finish together: round 4, lean kernel timeline); the last workgroup to
// finish (ticket in gacc[7]) publishes the totals to P.gstat and hands the accumulator back zeroed to the next launch.
// Every thread of the workgroup must call it; lane 0 of each wavefront carries that wavefront's values.
__device__ __forceinline__ void fold_status(const AdmmParams &P, float m0, float m1, float m2, float m3,
                                            int unsolved_in_wave, int tid) {
    __shared__ float s_fold_m[4][4];
    __shared__ int s_fold_u[4];
    const int w = tid >> 6, nw = ((int)blockDim.x + 63) >> 6;
    const bool merged = nw <= 4;
    if ((tid & 63) == 0) {
        if (merged) {
            s_fold_m[w][0] = m0, s_fold_m[w][1] = m1, s_fold_m[w][2] = m2, s_fold_m[w][3] = m3;
            s_fold_u[w] = unsolved_in_wave;
        } else {
            atomicMax(&P.gacc[0], __float_as_uint(m0));
            atomicMax(&P.gacc[1], __float_as_uint(m1));
            atomicMax(&P.gacc[2], __float_as_uint(m2));
            atomicMax(&P.gacc[3], __float_as_uint(m3));
            if (unsolved_in_wave) atomicAdd(&P.gacc[4], (uint32_t)unsolved_in_wave);
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (merged) {
            int un = 0;
            for (int i = 0; i < nw; ++i) {
                m0 = i ? fmaxf(m0, s_fold_m[i][0]) : s_fold_m[0][0];
                m1 = i ? fmaxf(m1, s_fold_m[i][1]) : s_fold_m[0][1];
                m2 = i ? fmaxf(m2, s_fold_m[i][2]) : s_fold_m[0][2];
                m3 = i ? fmaxf(m3, s_fold_m[i][3]) : s_fold_m[0][3];
                un += s_fold_u[i];
            }
            atomicMax(&P.gacc[0], __float_as_uint(m0));
            atomicMax(&P.gacc[1], __float_as_uint(m1));
            atomicMax(&P.gacc[2], __float_as_uint(m2));
            atomicMax(&P.gacc[3], __float_as_uint(m3));
            if (un) atomicAdd(&P.gacc[4], (uint32_t)un);
        }
        __threadfence();  // this workgroup's contributions before its ticket
        if (atomicAdd(&P.gacc[7], 1u) == gridDim.x - 1) {
            __threadfence();
#pragma unroll
            for (int i = 0; i < 5; ++i) P.gstat[i] = atomicExch(&P.gacc[i], 0u);
            atomicExch(&P.gacc[6], 0u);   // (persistent kernels' tile counter)
            atomicExch(&P.gacc[7], 0u);
        }
    }
}
#endif

}  // namespace tmpc
