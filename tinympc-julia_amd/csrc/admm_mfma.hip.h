// Fused ADMM kernel on the fp64 matrix cores: "mfma<nx,nu,N>", for one-shot solves of shapes with nx <= 12, nu <= 4.
//
// One instance's mat-vecs are too small for a matrix core, but a batch sharing one (A, B, Kinf, ...) is a dense
// contraction: 16 instances side by side make X (nx x 16), and x+ = A x + B u for all of them is a 16 x nx x 16
// product.  v_mfma_f64_16x16x4f64 (D = A B + C, A: 16 x 4, B: 4 x 16) has, on gfx950,
//     A operand: lane l holds A[l % 16][l / 16]          B operand: lane l holds B[l / 16][l % 16]
//     result   : lane l, register v holds D[4 v + l / 16][l % 16]          (experiments/mfma_probe.hip)
// so with lane l = 16 g + j working on instance j of its wavefront and owning state rows {g, 4 + g, 8 + g} (registers
// v = 0, 1, 2) and input row g (register v = 3), a RESULT IS ALREADY LAID OUT AS THE NEXT PRODUCT'S B OPERAND: register
// s of lane group g is row 4 s + g, i.e. K-slice s.  The whole forward / backward recurrence chains MFMAs without a
// single cross-lane move:
//     forward  : c = [B; 0] (-d);  c += [A - B Kinf; -Kinf] x (one MFMA per K-slice)  ->  c[0..2] = x+,  u = c[3] - d
//     backward : c = {q, r} + [-Kinf^T; 0] r;  c += [AmBKt; B^T] p              ->  c[0..2] = p-,  c[3] = B^T p + r
//                d = [0; Quu_inv] c[3], issued one knot later
// with the stacked matrices ([A; -Kinf] is 16 x 12 for the quadrotor: a full tile) spread over the 64 lanes, three
// doubles per lane and matrix, held in registers for the whole solve: no LDS coefficient traffic, no DPP broadcasts,
// and the 48 + 48 fp64 FMAs per lane and knot pair of the quad kernel leave the VALU, which does the elementwise
// slack / dual / cost work while the matrix core runs.
//
// Scope: plain solves (cold one-shot, warm-started, workspace-keeping, chunked with compaction); the fused closed loop
// stays on the quad kernel.  An instance that converges stores its solution (and, WS, its workspace) at that iteration
// and idles (its lanes keep iterating, results discarded) until its wavefront is done.
//
// WS (workspace variant): d, y, g, v, z are loaded from / saved to the persistent workspace.  The slack registers are
// updated in place, but a solve that converges must leave the PREVIOUS iteration's slack in v, z (admm.cpp:181-197
// returns before `v = vnew`), and that is what the next solve's first dual residual is measured against.  So on
// iterations that can converge each lane parks the slack it is about to overwrite in LDS ([row][64 instances],
// (nx N + nu (N-1)) * 256 B per workgroup: 119 KB for the quadrotor) and a converging instance saves v, z from there.
#pragma once
#include <hip/hip_runtime.h>

#include "admm_params.h"

namespace tmpc {

typedef double mf_d4 __attribute__((ext_vector_type(4)));

#ifndef TMPC_MFMA_TWO_WAVES
#define TMPC_MFMA_TWO_WAVES 1
#endif
#ifndef TMPC_MFMA_TWO_WAVES_MAX_STATE
#define TMPC_MFMA_TWO_WAVES_MAX_STATE 120  // floats of state per lane: N = 20 yes (85 dwords spilled, -3 %), N = 30 no (261 spilled, +-0)
#endif
#ifndef TMPC_MFMA_INTERLEAVE_B
#define TMPC_MFMA_INTERLEAVE_B 0  // same for the backward sweep (measured: no gain)
#endif
#ifndef TMPC_MFMA_INTERLEAVE
#define TMPC_MFMA_INTERLEAVE 6  // > 0: VALU instructions the scheduler is asked to place after each matrix product
#endif

template <int NX, int NU, int N>
struct MfmaShape {
    static_assert(NX >= 1 && NX <= 12 && NU >= 1 && NU <= 4, "mfma kernel: nx <= 12, nu <= 4");
    static constexpr int VX = (NX + 3) / 4;             // state rows (registers) per lane
    static constexpr int NF = 5 * VX + 3;               // operand doubles per lane: Mf[VX] Bf Mb[VX] KTn QI PT[VX] AT[VX] SP[VX]
    static constexpr int O_MF = 0, O_BF = VX, O_MB = VX + 1, O_KT = 2 * VX + 1, O_QI = 2 * VX + 2, O_PT = 2 * VX + 3,
                         O_AT = 3 * VX + 3,             // AT: [A^T; B^T] (adaptive rho: the norms' A'g, B'g)
                         O_SP = 4 * VX + 3;             // SP: (dPinf/drho)^T (adaptive rho: the terminal knot's Pinf_b x)
    // behind the lane fields: the family's Kinf, row-major [NU][NX] (adaptive rho: an instance's own Kinf enters as a
    // correction to the products formed with this one)
    // ... and its Pinf, row-major [NX][NX] (what an instance's adaptive state is rebuilt from when it finishes)
    // ... and its rho as a double (AdmmParams::rho is a float)
    static constexpr int O_K0 = NF * 64, O_P0 = O_K0 + NU * NX, O_RHO0 = O_P0 + NX * NX, COEF_DOUBLES = O_RHO0 + 1;
    // bounds pack (fp32): xmin[N][NX] xmax[N][NX] umin[N-1][NU] umax[N-1][NU] Qd[NX] Rd[NU]
    static constexpr int B_XMIN = 0, B_XMAX = N * NX, B_UMIN = 2 * N * NX, B_UMAX = 2 * N * NX + (N - 1) * NU,
                         B_QD = 2 * N * NX + 2 * (N - 1) * NU, B_RD = B_QD + NX, BOUNDS_LEN = B_RD + NU;
    static constexpr int REFS_LEN = N * NX + (N - 1) * NU;  // shared references: xref[N][NX] uref[N-1][NU]
};

template <int NX>
constexpr int VXof() {
    return (NX + 3) / 4;
}

template <int I, int E, class F>
__device__ __forceinline__ void mf_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        mf_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ mf_d4 mf_mma(double a, double b, mf_d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// max over the four lanes (16 apart) of an instance
__device__ __forceinline__ float mf_inst_max(float m) {
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    return m;
}

// ---- exchanges between the four lanes (16 apart) of an instance, on the VALU (adaptive rho) ----
// v_permlane32_swap exchanges the upper 32 lanes of its first operand with the lower 32 of its second, v_permlane16_swap
// the odd 16-lane rows of the first with the even rows of the second (inline asm: experiments/permlane_probe.hip; the
// s_nop covers the VALU-write -> permlane-read hazard).
__device__ __forceinline__ void mf_swap32(unsigned &a, unsigned &b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void mf_swap16(unsigned &a, unsigned &b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// reduce-scatter: every lane holds a partial sum for each of the instance's four lane groups; lane group g gets the total
// of component g.  swap32 on (pe0, pe2): the lower half keeps its pe0 and receives the upper half's pe0, the upper half
// keeps pe2 and receives pe2 — their sum is the pair total in both; likewise (pe1, pe3); then swap16 on the two totals.
__device__ __forceinline__ double mf_reduce_scatter4(double p0, double p1, double p2, double p3) {
    unsigned a0 = (unsigned)__double2loint(p0), a1 = (unsigned)__double2hiint(p0), b0 = (unsigned)__double2loint(p2), b1 = (unsigned)__double2hiint(p2);
    mf_swap32(a0, b0), mf_swap32(a1, b1);
    const double t0 = __hiloint2double((int)a1, (int)a0) + __hiloint2double((int)b1, (int)b0);
    unsigned c0 = (unsigned)__double2loint(p1), c1 = (unsigned)__double2hiint(p1), d0 = (unsigned)__double2loint(p3), d1 = (unsigned)__double2hiint(p3);
    mf_swap32(c0, d0), mf_swap32(c1, d1);
    const double t1 = __hiloint2double((int)c1, (int)c0) + __hiloint2double((int)d1, (int)d0);
    unsigned e0 = (unsigned)__double2loint(t0), e1 = (unsigned)__double2hiint(t0), f0 = (unsigned)__double2loint(t1), f1 = (unsigned)__double2hiint(t1);
    mf_swap16(e0, f0), mf_swap16(e1, f1);
    return __hiloint2double((int)e1, (int)e0) + __hiloint2double((int)f1, (int)f0);
}
// all-gather of a float: out[a] = the value lane group a holds, in every lane of the instance
__device__ __forceinline__ void mf_all_gather4(float v, float (&out)[4]) {
    unsigned a = __float_as_uint(v), b = a;
    mf_swap32(a, b);                       // (a, b) = (value of group g % 2, value of group g % 2 + 2) in every lane
    unsigned p = a, q = a, r = b, t = b;
    mf_swap16(p, q), mf_swap16(r, t);      // (p, q) = (group 0, group 1), (r, t) = (group 2, group 3)
    out[0] = __uint_as_float(p), out[1] = __uint_as_float(q), out[2] = __uint_as_float(r), out[3] = __uint_as_float(t);
}
__device__ __forceinline__ double mf_inst_max_d(double m) {
    m = fmax(m, __shfl_xor(m, 16, 64));
    m = fmax(m, __shfl_xor(m, 32, 64));
    return m;
}

// dynamic LDS of the WS variant: the parked slack of the workgroup's 64 instances
template <int NX, int NU, int N>
constexpr size_t mfma_ws_lds_bytes() {
    return (size_t)(NX * N + NU * (N - 1)) * 64 * sizeof(float);
}

// Workgroups per CU the register allocation is held to: 2 (two wavefronts per SIMD, 256 registers each, so one
// wavefront's VALU work and result latencies hide behind the other's matrix products) where the state is small enough
// to fit without spilling, else 1.
template <int NX, int NU, int N, int REFS, bool XB, bool WS>
constexpr int mfma_blocks_per_cu() {
    return (TMPC_MFMA_TWO_WAVES && VXof<NX>() * N + 3 * (N - 1) <= TMPC_MFMA_TWO_WAVES_MAX_STATE && !XB && !WS && REFS != REF_PER_INSTANCE) ? 2 : 1;
}

// RF ("refill"): tolerance-terminated one-shot solves of a batch larger than the chip holds at once.  The launch has as
// many workgroups as are resident together; an instance slot (16 per wavefront) whose instance has finished — converged
// at a check, or out of iterations — stores its solution and takes the next unstarted instance off a global counter,
// cold: the iteration sequence of every instance is exactly the plain kernel's, but no slot idles behind the slowest
// instance of its wavefront (config 5's shard: mean 46 of 100 iterations, yet nearly every wavefront holds an instance
// that runs to max_iter).  Needs max_iter % check_termination == 0, so that a slot is only ever refilled on a check
// iteration and every instance's own check schedule coincides with the launch's.
//
// ADP: adaptive rho (admm.cpp:147-174 with rho_benchmark.cpp:44-213) — what the reference builds for exactly this shape
// (its sensitivity tables are the quadrotor's, tiny_api.cpp:269-329).  Every instance carries its own rho, Kinf and Pinf
// (the solver's adaptive state, AdmmParams::adapt), re-predicted on the iterations i > 0, i % 5 == 0 from norms gathered
// during that iteration's forward sweep; AmBKt and Quu_inv stay the family's, as in the reference.  A per-instance Kinf
// does not fit a product whose matrix operand is shared by the 16 instances, so it enters as a correction: with
// dK = Kinf_b - Kinf_0 (12 doubles per lane: dK[:, the lane's three state rows])
//     forward : d' = d + dK x_k (partial products in every lane, a reduce-scatter over the instance's four lanes), then the
//               same products with d' — x+ = (A - B Kinf_0) x - B d' = A x + B u, u = -Kinf_0 x - d' = -Kinf_b x - d.  The
//               product with d' now closes the chain instead of opening it.
//     backward: -dK' r is added to the accumulator's start on the VALU (r gathered over the instance's lanes; off the chain).
// The updates are first-order in rho with constant tables, so an instance's matrices are the family's plus
// (rho_b - rho_family) x table (the solver keeps the adaptive state in that form: Solver::adapt_pure): dK comes from rho_b
// alone, Pinf_b — only needed at the terminal knot: its norm rows on adapting iterations, the reference term when there are
// references — is two products (Pinf_0', dPinf') per use, and the adaptive state in HBM is read for rho_b at entry and
// written (rho, Kinf, Pinf) once, when the instance finishes.
// The norms of rho_benchmark.cpp are gathered as in the quad kernel's ADP variant (admm_quad.hip.h: rows of knot k - 1 at
// knot k, the terminal knot's own), A'g / B'g by three more products with [A'; B'].
template <int NX, int NU, int N, int REFS, bool XB, bool WS = false, bool RF = false, bool ADP = false>
__global__ __launch_bounds__(256, (ADP ? 1 : mfma_blocks_per_cu<NX, NU, N, REFS, XB, WS>())) void admm_mfma_kernel(const AdmmParams P) {
    static_assert(!RF || (!WS && REFS != REF_PER_INSTANCE), "refill: one-shot solves, shared or zero references");
    static_assert(!ADP || !RF, "adaptive rho: not with refill");
    // XB = false with WS: the caller guarantees that the workspace's state dual is zero and stays zero (no finite
    // state bound now, none since the last reset) — g is then neither loaded, carried nor written.
    using S = MfmaShape<NX, NU, N>;
    constexpr int VX = S::VX, T = 256;
    __shared__ float s_bnd[S::BOUNDS_LEN];
    __shared__ float s_ref[REFS == REF_SHARED ? S::REFS_LEN : 1];
    extern __shared__ float s_old[];  // WS: [row of v | row of z][64 instances]
    const int tid = threadIdx.x;
    for (int i = tid; i < S::BOUNDS_LEN; i += T) s_bnd[i] = P.bounds[i];
    if constexpr (REFS == REF_SHARED)
        for (int i = tid; i < S::REFS_LEN; i += T) s_ref[i] = i < N * NX ? P.xref[i] : P.uref[i - N * NX];
    __syncthreads();

    const int l = tid & 63, g = l >> 4, j = l & 15;
    const int inst = (tid >> 6) * 16 + j;  // instance of the workgroup
    const long slot = (long)blockIdx.x * 64 + inst;
    bool active = slot < P.batch;                                // (RF: the slot still has an instance)
    long b = (active && P.idx) ? P.idx[slot] : slot;             // (RF: the instance the slot is working on)
    int it0 = 0;                                                 // RF: the launch's iteration count when the slot's instance started
    float fm0 = 0.f, fm1 = 0.f, fm2 = 0.f, fm3 = 0.f;            // RF: residual maxima / unsolved count over the slot's finished instances
    int f_unsolved = 0;
    constexpr long EX = (long)NX * N, EU = (long)NU * (N - 1);
    const bool uok = g < NU;
    bool xok[VX];
#pragma unroll
    for (int v = 0; v < VX; ++v) xok[v] = 4 * v + g < NX;

    // matrix operands of this lane (fp64, [field][64 lanes] in HBM), constant for the solve
    const double *gc = reinterpret_cast<const double *>(P.coef);
    double cf[S::NF];
#pragma unroll
    for (int f = 0; f < S::NF; ++f) cf[f] = gc[f * 64 + l];
    float qd[VX], rd = uok ? s_bnd[S::B_RD + g] : 0.f;
#pragma unroll
    for (int v = 0; v < VX; ++v) qd[v] = xok[v] ? s_bnd[S::B_QD + 4 * v + g] : 0.f;
    float rho = P.rho;            // (ADP: this instance's own, re-predicted every 5th iteration)
    double rho_d = (double)P.rho;

    // per-instance state of this lane: its rows of the state dual, of vnew (in place of v), and its input row's
    // y, znew (in place of z), d.  No state dual without an active state bound (identically zero in a cold one-shot solve).
    float sg[XB ? N : 1][VX], sw[N][VX], sy[N - 1], szw[N - 1], sd[N - 1];
    float xr[REFS == REF_PER_INSTANCE ? N : 1][VX], ur[REFS == REF_PER_INSTANCE ? N - 1 : 1];
    double x0[VX];
#pragma unroll
    for (int v = 0; v < VX; ++v)
        x0[v] = (active && xok[v]) ? (P.x0d ? P.x0d[b * NX + 4 * v + g] : (double)P.x0[b * NX + 4 * v + g]) : 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int v = 0; v < VX; ++v) {
            if constexpr (XB) sg[k][v] = 0.f;
            sw[k][v] = 0.f;
            if constexpr (REFS == REF_PER_INSTANCE)
                xr[k][v] = (active && xok[v]) ? P.xref[b * EX + k * NX + 4 * v + g] : 0.f;
        }
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        sy[k] = szw[k] = sd[k] = 0.f;
        if constexpr (REFS == REF_PER_INSTANCE) ur[k] = (active && uok) ? P.uref[b * EU + k * NU + g] : 0.f;
    }
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    if constexpr (WS) {
        if (!P.cold_start && active) {  // warm start: d, y, g, v, z of the previous solve
#pragma unroll
            for (int v = 0; v < VX; ++v)
                if (xok[v]) {
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        if constexpr (XB) sg[k][v] = P.sg[b * EX + k * NX + 4 * v + g];
                        sw[k][v] = P.sv[b * EX + k * NX + 4 * v + g];
                    }
                }
            if (uok) {
#pragma unroll
                for (int k = 0; k < N - 1; ++k) {
                    sy[k] = P.sy[b * EU + k * NU + g];
                    szw[k] = P.sz[b * EU + k * NU + g];
                    sd[k] = P.sd[b * EU + k * NU + g];
                }
            }
            res0 = P.res[b * 4 + 0], res1 = P.res[b * 4 + 1], res2 = P.res[b * 4 + 2], res3 = P.res[b * 4 + 3];
        }
    }
    // ---- ADP: the instance's adaptive state (rho, Kinf as dK against the family's, the terminal reference term) ----
    constexpr int VA = ADP ? VX : 1;
    double dk[4][VA], accP[VA], spx[VA];
    const long AB = ADP ? P.adapt_stride : 0;
    double *const ad = ADP ? P.adapt + b : nullptr;
    if constexpr (ADP) {
        if (active) rho_d = ad[0], rho = (float)rho_d;
#pragma unroll
        for (int v = 0; v < VX; ++v) {
            accP[v] = spx[v] = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                dk[a][v] = (active && xok[v] && a < NU) ? (rho_d - gc[S::O_RHO0]) * P.sens[a + (4 * v + g) * NU] : 0.0;
        }
    }
    const double rho_fam = ADP ? gc[S::O_RHO0] : 0.0, rho_entry = rho_d;

    auto ref_x = [&](auto kk, int v) -> float {
        constexpr int K = decltype(kk)::value;
        if constexpr (REFS == REF_SHARED) return xok[v] ? s_ref[K * NX + 4 * v + g] : 0.f;
        else if constexpr (REFS == REF_PER_INSTANCE) return xr[K][v];
        else return 0.f;
    };
    auto ref_u = [&](auto kk) -> float {
        constexpr int K = decltype(kk)::value;
        if constexpr (REFS == REF_SHARED) return uok ? s_ref[N * NX + K * NU + g] : 0.f;
        else if constexpr (REFS == REF_PER_INSTANCE) return ur[K];
        else return 0.f;
    };
    constexpr float kInf = __builtin_inff();
    if constexpr (ADP && REFS != REF_ZERO) {
        // the terminal reference term Pinf_b' xref_{N-1} of this lane's rows, and what a unit step of rho adds to it (on
        // the VALU from the pack's row-major Pinf_0 and the table: once per launch; more products here have crashed the
        // compiler's AGPR-copy rewrite on some horizons)
        const double *sP = P.sens + NU * NX, *P0 = gc + S::O_P0;
#pragma unroll
        for (int v = 0; v < VX; ++v)
            if (active && xok[v]) {
                const int r = 4 * v + g;
                double a0 = 0.0, a1 = 0.0;
                for (int jj = 0; jj < NX; ++jj) {
                    const double xrj = REFS == REF_SHARED ? (double)s_ref[(N - 1) * NX + jj] : (double)P.xref[b * EX + (N - 1) * NX + jj];
                    a0 = fma(P0[jj * NX + r], xrj, a0);
                    a1 = fma(sP[jj + r * NX], xrj, a1);
                }
                spx[v] = a1, accP[v] = a0 + (rho_d - rho_fam) * a1;
            }
    }

    int it = 0, conv = 0;
    const int ct = P.check_termination;
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;

    auto store_solution = [&]() {  // this instance's solution and status: vnew / znew of the iteration just run
#pragma unroll
        for (int v = 0; v < VX; ++v)
            if (xok[v]) {
#pragma unroll
                for (int k = 0; k < N; ++k) P.xout[b * EX + k * NX + 4 * v + g] = sw[k][v];
            }
        if (uok) {
#pragma unroll
            for (int k = 0; k < N - 1; ++k) P.uout[b * EU + k * NU + g] = szw[k];
        }
        if (g == 0) {
            P.iter[b] = P.iter_offset + it - it0;
            P.solved[b] = conv;
            P.res[b * 4 + 0] = res0;
            P.res[b * 4 + 1] = res1;
            P.res[b * 4 + 2] = res2;
            P.res[b * 4 + 3] = res3;
        }
    };

    // WS: the workspace as the reference leaves it.  `parked`: v, z are the slack of the iteration BEFORE the one just
    // run (a converged solve), read back from where this lane parked them.
    auto store_workspace = [&](bool parked) {
#pragma unroll
        for (int v = 0; v < VX; ++v)
            if (xok[v]) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    if constexpr (XB) P.sg[b * EX + k * NX + 4 * v + g] = sg[XB ? k : 0][v];
                    P.sv[b * EX + k * NX + 4 * v + g] = parked ? s_old[(k * NX + 4 * v + g) * 64 + inst] : sw[k][v];
                }
            }
        if (uok) {
#pragma unroll
            for (int k = 0; k < N - 1; ++k) {
                P.sy[b * EU + k * NU + g] = sy[k];
                P.sz[b * EU + k * NU + g] = parked ? s_old[(N * NX + k * NU + g) * 64 + inst] : szw[k];
                P.sd[b * EU + k * NU + g] = sd[k];
            }
        }
    };

    // ADP: the instance's adaptive state as the reference leaves it (rho, Kinf, Pinf: admm.cpp:160-172 accumulated)
    auto store_adapt = [&]() {
        if constexpr (ADP) {
            if (rho_d != rho_entry) {
                const double dr = rho_d - rho_fam;
                const double *sK = P.sens, *sP = P.sens + NU * NX, *K0 = gc + S::O_K0, *P0 = gc + S::O_P0;
#pragma unroll
                for (int v = 0; v < VX; ++v)
                    if (xok[v]) {
                        const int r = 4 * v + g;
#pragma unroll
                        for (int a = 0; a < NU; ++a) ad[(long)(1 + a + r * NU) * AB] = K0[a * NX + r] + dr * sK[a + r * NU];
                        for (int jj = 0; jj < NX; ++jj) ad[(long)(1 + NU * NX + jj + r * NX) * AB] = P0[jj * NX + r] + dr * sP[jj + r * NX];
                    }
                if (g == 0) ad[0] = rho_d;
            }
        }
    };

    for (int i = 0; RF || i < P.max_iter; ++i) {
        const bool check = ct > 0 && (i + 1) % ct == 0;                       // every lane's it == i here (RF: it - it0 is a multiple of ct behind)
        const bool need_res = check && (can_converge || i + 1 == last_check_it);
        // ================= fused forward sweep (admm.cpp:25-69, :93-96) =================
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
        double x[VX], c3_pend = 0.0, nd_pend = 0.0;
#pragma unroll
        for (int v = 0; v < VX; ++v) x[v] = x0[v];
        // ADP: the iterations that adapt (admm.cpp:147, the loop index before it is bumped) gather the norms on the way
        const bool adapt_now = ADP && i > 0 && i % 5 == 0;
        const float rho_lin = rho;   // the linear cost of this iteration is formed before the adaptation (admm.cpp:139 vs :147)
        // (|P x| and |q| take the same values on every row but the terminal knot's, |A x - z| and |z| the same on the state rows:
        // one running maximum each for the shared part — a_pq, a_vnm — a third of the maxima less per knot)
        double a_pri = 0.0, a_axm = 0.0, a_zm = 0.0, a_dres = 0.0, a_pxm = 0.0, a_atym = 0.0, a_qm = 0.0, a_pq = 0.0, a_vnm = 0.0;
        double a_xp[VA], a_up = 0.0, a_yp = 0.0, accP_new[VA];
        float a_gp[VA];
#pragma unroll
        for (int v = 0; v < VA; ++v) a_xp[v] = 0.0, a_gp[v] = 0.f, accP_new[v] = accP[v];
        auto upmax = [](double &m, double v) { m = __builtin_fmax(m, __builtin_fabs(v)); };   // (one v_max_f64 with |.| on its operand)
        mf_for<0, N>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            asm volatile("" ::: "memory");  // LDS constants (bounds, shared references) are re-read per knot, not hoisted
            // x+ = A x + B u, u = -Kinf x - d, regrouped as ONE accumulation chain that never waits for u and whose first
            // product does not depend on x at all (the matrix core runs it while the previous knot's result drains):
            //   c = [B; 0] (-d);   c += [A - B Kinf; -Kinf] x      ->  c[0..2] = x+,  c[3] = -Kinf x  (u = c[3] - d)
            // fp64 throughout: the regrouping moves results by a few 1e-16.  The chain starts from the constant-zero
            // accumulator (an inline operand), so no register tuple is initialised or copied between knots.
            mf_d4 c = {0.0, 0.0, 0.0, 0.0};
            double nd = 0.0;
            if constexpr (k < N - 1 && !ADP) {
                nd = -(double)sd[k];
                c = mf_mma(cf[S::O_BF], nd, c);                                // [B; 0] (-d)
                mf_for<0, VX>([&](auto ss) {
                    constexpr int s = decltype(ss)::value;
                    c = mf_mma(cf[S::O_MF + s], x[s], c);                      // + [A - B Kinf; -Kinf] x
                });
            }
            if constexpr (k < N - 1 && ADP) {
                // this instance's own Kinf: d' = d + dK x_k, summed over the instance's lanes while the x products run
                mf_for<0, VX>([&](auto ss) {
                    constexpr int s = decltype(ss)::value;
                    c = mf_mma(cf[S::O_MF + s], x[s], c);                      // [A - B Kinf_0; -Kinf_0] x
                });
                double pe[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int a = 0; a < NU; ++a)
#pragma unroll
                    for (int v = 0; v < VX; ++v) pe[a] = fma(dk[a][v], x[v], pe[a]);
                nd = -(double)sd[k] - mf_reduce_scatter4(pe[0], pe[1], pe[2], pe[3]);
                c = mf_mma(cf[S::O_BF], nd, c);                                // + [B; 0] (-d')
            }
            // slack / dual of the state rows at this knot (the matrix core works on the products meanwhile)
            float a_xf[VA], a_vn[VA], a_gn[VA];                               // ADP: this knot's x (as the sets see it), slack, new dual
#pragma unroll
            for (int v = 0; v < VX; ++v) {
                const float xf = (float)x[v];
                const float gk = XB ? sg[XB ? k : 0][v] : 0.f;
                float vn = XB ? xf + gk : xf;
                if constexpr (XB) {
                    const float lo = xok[v] ? s_bnd[S::B_XMIN + k * NX + 4 * v + g] : -kInf;
                    const float hi = xok[v] ? s_bnd[S::B_XMAX + k * NX + 4 * v + g] : kInf;
                    vn = fminf(hi, fmaxf(lo, vn));
                    sg[XB ? k : 0][v] = (gk + xf) - vn;
                }
                if constexpr (ADP) a_xf[v] = xf, a_vn[v] = vn, a_gn[v] = XB ? (gk + xf) - vn : 0.f;
                if (need_res) {
                    pri_x = fmaxf(pri_x, fabsf(xf - vn));
                    dua_x = fmaxf(dua_x, fabsf(sw[k][v] - vn));
                }
                if constexpr (WS)
                    if (xok[v]) s_old[(k * NX + 4 * v + g) * 64 + inst] = sw[k][v];  // unconditional: no branch in the knot
                sw[k][v] = vn;
            }
            // slack / dual of the input row of the knot BEFORE (its u left the matrix core while this knot's chain was being
            // issued): by then nothing here waits for a result
            if constexpr (k >= 1) {
                constexpr int j = k - 1;
                const double u = c3_pend + nd_pend;                            // -Kinf x - d
                const float uf = (float)u, yk = sy[j];
                float zn = uf + yk;
                const float lo = uok ? s_bnd[S::B_UMIN + j * NU + g] : -kInf, hi = uok ? s_bnd[S::B_UMAX + j * NU + g] : kInf;
                zn = fminf(hi, fmaxf(lo, zn));
                sy[j] = (yk + uf) - zn;
                if (need_res) {
                    pri_u = fmaxf(pri_u, fabsf(uf - zn));
                    dua_u = fmaxf(dua_u, fabsf(szw[j] - zn));
                }
                if constexpr (WS)
                    if (uok) s_old[(N * NX + j * NU + g) * 64 + inst] = szw[j];
                szw[j] = zn;
                if constexpr (ADP) {
                    if (adapt_now) {
                        asm volatile("" ::: "memory");
                        upmax(a_pri, (double)uf - (double)zn);
                        upmax(a_axm, (double)uf);
                        upmax(a_zm, (double)zn);
                        a_up = (double)uf, a_yp = (double)((yk + uf) - zn);
                    }
                }
            }
            if constexpr (ADP) {
                if (adapt_now) {   // norm rows of knot k - 1 (they needed g_k), the terminal knot's own (admm_quad.hip.h, admm_streamg.hip.h)
                    // (the block has no side effects, and left alone the compiler runs it — three products and ~60 fp64
                    // instructions per knot — on EVERY iteration and selects at the end)
                    asm volatile("" ::: "memory");
                    if constexpr (k >= 1) {
                        double atx[VX], btg = 0.0;
#pragma unroll
                        for (int v = 0; v < VX; ++v) atx[v] = 0.0;
                        if constexpr (XB) {
                            mf_d4 t = {0.0, 0.0, 0.0, 0.0};
                            mf_for<0, VX>([&](auto ss) {
                                constexpr int s = decltype(ss)::value;
                                t = mf_mma(cf[S::O_AT + s], (double)a_gn[s], t);   // [A'; B'] g_k
                            });
#pragma unroll
                            for (int v = 0; v < VX; ++v) atx[v] = t[v] - (k >= 2 ? (double)a_gp[v] : 0.0);
                            btg = t[3];
                        }
#pragma unroll
                        for (int v = 0; v < VX; ++v) {
                            const double qv = (double)qd[v] * a_xp[v];
                            upmax(a_dres, qv + qv + atx[v]);
                            upmax(a_pq, qv);                 // |P x| and |q| of the row
                            upmax(a_atym, atx[v]);
                            upmax(a_vnm, (double)a_vn[v]);   // |A x - z| and |z|: A x + B u - x_k vanishes against the rollout's own x_k
                        }
                        const double px = (double)rd * a_up, aty = a_yp + btg;
                        upmax(a_dres, px + px + aty);
                        upmax(a_pq, px);
                        upmax(a_atym, aty);
                    }
                    if constexpr (k == N - 1) {   // Pinf_b x + Q~ x - g of the terminal knot: Pinf_b = Pinf_0 + (rho_b - rho_family) dPinf
                        mf_d4 p0x = {0.0, 0.0, 0.0, 0.0}, spx4 = {0.0, 0.0, 0.0, 0.0};
                        mf_for<0, VX>([&](auto ss) {
                            constexpr int s2 = decltype(ss)::value;
                            p0x = mf_mma(cf[S::O_PT + s2], (double)a_xf[s2], p0x);
                            spx4 = mf_mma(cf[S::O_SP + s2], (double)a_xf[s2], spx4);
                        });
#pragma unroll
                        for (int v = 0; v < VX; ++v) {
                            const double px = p0x[v] + (rho_d - rho_fam) * spx4[v];
                            const double qv = (double)qd[v] * (double)a_xf[v], aty = -(double)a_gn[v];
                            upmax(a_dres, px + qv + aty);
                            upmax(a_pxm, px);
                            upmax(a_qm, qv);
                            upmax(a_atym, aty);
                        }
                    }
#pragma unroll
                    for (int v = 0; v < VX; ++v) a_xp[v] = (double)a_xf[v], a_gp[v] = a_gn[v];
                }
            }
            if constexpr (k < N - 1) {
                c3_pend = c[3];
                nd_pend = nd;
#pragma unroll
                for (int v = 0; v < VX; ++v) x[v] = c[v];
                if constexpr (TMPC_MFMA_INTERLEAVE > 0) {  // one matrix product, then a share of the knot's VALU work, ...
#pragma unroll
                    for (int q4 = 0; q4 < 1 + VX; ++q4) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, TMPC_MFMA_INTERLEAVE, 0);
                    }
                }
            }
        });
        it += 1;
        if constexpr (ADP) {
            if (adapt_now) {
                // predict_rho (rho_benchmark.cpp:173-195), then the first-order update of Kinf, Pinf (admm.cpp:160-172)
                const double pri = mf_inst_max_d(fmax(a_pri, a_vnm)), axm = mf_inst_max_d(a_axm), zm = mf_inst_max_d(fmax(a_zm, a_vnm)),
                             dres = mf_inst_max_d(a_dres), pxm = mf_inst_max_d(fmax(a_pxm, a_pq)), atym = mf_inst_max_d(a_atym),
                             qm = mf_inst_max_d(fmax(a_qm, a_pq));
                if (active && !conv) {   // (a finished instance idles: its adaptive state is what it finished with)
                    const double eps = 1e-10, prin = axm > zm ? axm : zm;
                    double duan = pxm > atym ? pxm : atym;
                    duan = qm > duan ? qm : duan;
                    const double ratio = (pri / (prin + eps)) / (dres / (duan + eps) + eps);
                    double nrho = rho_d * sqrt(ratio);
                    if (P.rho_clip) nrho = nrho < (double)P.rho_min ? (double)P.rho_min : (nrho > (double)P.rho_max ? (double)P.rho_max : nrho);
                    const double delta = nrho - rho_d;
                    const double *sK = P.sens;
#pragma unroll
                    for (int v = 0; v < VX; ++v)
                        if (xok[v]) {
                            const int r = 4 * v + g;
#pragma unroll
                            for (int a = 0; a < NU; ++a) dk[a][v] += delta * sK[a + r * NU];   // column r of Kinf, as this lane's correction
                            accP_new[v] = accP[v] + delta * spx[v];
                        }
                    rho_d = nrho;
                    rho = (float)nrho;
                }
            }
        }
        bool newly = false;
        if (need_res) {
            const float r0 = mf_inst_max(pri_x), r1 = mf_inst_max(dua_x) * rho, r2 = mf_inst_max(pri_u),
                        r3 = mf_inst_max(dua_u) * rho;
            if (!conv) {  // a finished instance keeps the residuals it finished with
                res0 = r0, res1 = r1, res2 = r2, res3 = r3;
                if (res0 < P.abs_pri_tol && res2 < P.abs_pri_tol && res1 < P.abs_dua_tol && res3 < P.abs_dua_tol) {
                    conv = 1;
                    newly = true;
                }
            }
        }
        bool fresh = false;                                                   // RF: the slot took a new instance in this iteration
        if constexpr (RF) {
            const bool fin = active && check && (newly || it - it0 >= P.max_iter);
            if (__builtin_amdgcn_ballot_w64(fin)) {
                long nb = -1;
                if (fin) {
                    store_solution();
                    fm0 = fmaxf(fm0, res0), fm1 = fmaxf(fm1, res1), fm2 = fmaxf(fm2, res2), fm3 = fmaxf(fm3, res3);
                    if (g == 0) {
                        f_unsolved += conv ? 0 : 1;
                        nb = 64L * gridDim.x + (long)atomicAdd(&P.gacc[6], 1u);   // the next unstarted instance
                    }
                }
                nb = __shfl(nb, j, 64);                                        // lane j (g = 0) holds the instance's draw
                if (fin) {
                    if (nb < (long)P.batch) {
                        fresh = true;
                        b = nb, it0 = it, conv = 0;
                        res0 = res1 = res2 = res3 = 0.f;
#pragma unroll
                        for (int v = 0; v < VX; ++v) x0[v] = xok[v] ? (double)P.x0[b * NX + 4 * v + g] : 0.0;
#pragma unroll
                        for (int k = 0; k < N; ++k)
#pragma unroll
                            for (int v = 0; v < VX; ++v) {
                                if constexpr (XB) sg[k][v] = 0.f;
                                sw[k][v] = 0.f;
                            }
#pragma unroll
                        for (int k = 0; k < N - 1; ++k) sy[k] = szw[k] = sd[k] = 0.f;
                    } else {
                        active = false;
                        conv = 1;
                    }
                }
            }
            if (!__builtin_amdgcn_ballot_w64(active)) break;
        } else {
            if (__builtin_amdgcn_ballot_w64(newly)) {
                if (newly && active) {
                    store_solution();
                    store_adapt();
                    if constexpr (WS)
                        if (P.save_state) store_workspace(true);
                }
            }
            if (!__builtin_amdgcn_ballot_w64(active && !conv)) break;
        }
        // ================= fused backward sweep (admm.cpp:75-83, :13-20) =================
        double p[VX];
        {
            mf_d4 c = {0.0, 0.0, 0.0, 0.0};
            if constexpr (REFS != REF_ZERO && !ADP) {
                mf_for<0, VX>([&](auto ss) {
                    constexpr int s = decltype(ss)::value;
                    c = mf_mma(cf[S::O_PT + s], (double)ref_x(std::integral_constant<int, N - 1>{}, s), c);  // Pinf^T xref
                });
            }
            if constexpr (ADP) {                                               // the instance's own Pinf, as of this iteration's linear cost
#pragma unroll
                for (int v = 0; v < VX; ++v) c[v] = accP[v], accP[v] = accP_new[v];
            }
#pragma unroll
            for (int v = 0; v < VX; ++v) p[v] = -c[v] - (double)(rho_lin * (sw[N - 1][v] - (XB ? sg[XB ? N - 1 : 0][v] : 0.f)));
        }
        // p- = q + AmBKt p - Kinf^T r and d = Quu_inv (B^T p + r), again ordered for the matrix core: the product that
        // does not depend on p opens the chain, and the d product of a knot is issued one knot later, when its operand
        // has long drained — the chain p -> p- is the only thing the sweep ever waits for.
        double t_pend = 0.0;                                                   // B^T p + r of the knot above
        mf_for<0, N - 1>([&](auto kk) {
            constexpr int k = N - 2 - decltype(kk)::value;
            constexpr std::integral_constant<int, k> kc{};
            asm volatile("" ::: "memory");
            const float rf = -(ref_u(kc) * rd) - rho_lin * (szw[k] - sy[k]);
            const double r = (double)rf;
            mf_d4 c;
#pragma unroll
            for (int v = 0; v < 3; ++v)
                c[v] = v < VX ? (double)(-(ref_x(kc, v < VX ? v : 0) * qd[v < VX ? v : 0]) -
                                         rho_lin * (sw[k][v < VX ? v : 0] - (XB ? sg[XB ? k : 0][v < VX ? v : 0] : 0.f)))
                              : 0.0;
            c[3] = r;
            if constexpr (ADP) {                                               // - dK' r: the instance's own Kinf in - Kinf' r
                float ra[4];
                mf_all_gather4(rf, ra);
#pragma unroll
                for (int a = 0; a < NU; ++a)
#pragma unroll
                    for (int v = 0; v < VX; ++v) c[v] = fma(-dk[a][v], (double)ra[a], c[v]);
            }
            c = mf_mma(cf[S::O_KT], r, c);                                     // {q, r} + [-Kinf^T; 0] r
            if constexpr (k < N - 2) {
                mf_d4 dq = {0.0, 0.0, 0.0, 0.0};
                dq = mf_mma(cf[S::O_QI], t_pend, dq);                          // [0; Quu_inv] (B^T p + r) of knot k + 1
                sd[k + 1] = (float)dq[3];
            }
            mf_for<0, VX>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                c = mf_mma(cf[S::O_MB + s], p[s], c);                          // + [AmBKt; B^T] p
            });
            t_pend = c[3];
#pragma unroll
            for (int v = 0; v < VX; ++v) p[v] = c[v];
            if constexpr (TMPC_MFMA_INTERLEAVE_B > 0) {
#pragma unroll
                for (int q5 = 0; q5 < 2 + VX; ++q5) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x002, TMPC_MFMA_INTERLEAVE_B, 1);
                }
            }
        });
        {
            mf_d4 dq = {0.0, 0.0, 0.0, 0.0};
            dq = mf_mma(cf[S::O_QI], t_pend, dq);
            sd[0] = (float)dq[3];
        }
        if constexpr (RF) {
            if (fresh) {   // a fresh instance has not run a forward sweep yet: its feed-forward term stays the cold start's zero
#pragma unroll
                for (int k = 0; k < N - 1; ++k) sd[k] = 0.f;
            }
        }
    }

    if constexpr (!RF) {
        if (active && !conv) {
            store_solution();
            store_adapt();
            if constexpr (WS)
                if (P.save_state) store_workspace(false);
        }
    }
    if constexpr (RF) {
        float m0 = fm0, m1 = fm1, m2 = fm2, m3 = fm3;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
            m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
        }
        int un = f_unsolved;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) un += __shfl_xor(un, off, 64);
        fold_status(P, m0, m1, m2, m3, un, tid);
    } else {
        float m0 = active ? res0 : 0.f, m1 = active ? res1 : 0.f, m2 = active ? res2 : 0.f, m3 = active ? res3 : 0.f;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {  // over the 16 instances of the wavefront (lanes of a group)
            m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
            m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
        }
        const unsigned long long unsolved = __builtin_amdgcn_ballot_w64(active && !conv && g == 0);
        fold_status(P, m0, m1, m2, m3, __popcll(unsolved), tid);
    }
}

}  // namespace tmpc
