// Fused ADMM kernel, recurrences on the fp64 matrix cores, constraint sets in a TRANSPOSED lane layout, the whole iterated
// state of a solve on chip: "mfmat<nx,nu,N>" — compile-time horizon, box bounds + affine dynamics term + one second-order
// cone per side (BASELINE config 4: the rocket, N = 50), cold one-shot solves AND the reference's default calling
// pattern: the workspace persists between solves (admm.cpp:111-115 resets counters only), per-instance early exit with the
// converged-exit quirk (admm.cpp:181-197), chunked solves, and the fused closed loop of
// examples/rocket_landing_constraints.jl:97-134 (set_x0 -> shifted references -> solve -> x+ = A x + B u0 + f).
//
// What was wrong with the three-wavefront kernels (admm_mfmar.hip.h, admm_mfmac.hip.h; profiles/r02_rocket_soc_*): the
// sets (slack / dual of every row, admm.cpp:43-69) ran in the matrix products' lane layout — lane = (row, instance), one
// knot at a time — so a cone needed cross-lane sums, its scalar part was computed four times, 25-50 % of the lanes had no
// row, and every knot cost ~150 VALU instructions per 16 instances (VALU issue 0.43 of the kernel, matrix cores 0.28).
// The previous slack (what the dual residual compares against, admm.cpp:95-96) had no place on chip and went through HBM
// around every check (6.5 x the algorithmic bytes at one check per solve, 27 GB per launch with the check live).
//
// Here ONE wavefront owns a tile of 16 instances and alternates between two lane layouts over the same LDS cells
//   cells[position p][row][instance]   rows 0..nx-1: x_p on the way to the sets, then sum over sets of (slack - dual) on the
//                                      way to the backward sweep; rows nx..: u_p, then the sets' sum, then t_p = B'p + r
//   (9 floats per knot for the rocket: 28.8 KB per tile at N = 50, four tiles per CU)
// * matrix layout (rollout admm.cpp:25-35, backward sweep admm.cpp:13-20,75-83): lane 16 g + j = rows g, 4 + g of x and
//   row g of u of instance j — the operand layout of v_mfma_f64_16x16x4f64, a result is the next product's operand
//   (admm_mfmac.hip.h has the algebra: M = [A - B Kinf, -B Quu_inv; -Kinf, -Quu_inv] applied to [x; t]);
// * sets layout (admm.cpp:43-69, 89-107): lane 16 q + j = knot 4 m + q of instance j, ALL rows of that knot in the lane's
//   registers: a cone's norm is a few in-lane FMAs, nothing is computed twice, every lane has work (13 groups m for 50
//   knots), ~40 VALU instructions per knot and 16 instances with the residuals, ~27 without.
// Between the layouts the data crosses through the cells (a transpose for free: both layouts address cell (p, row, j)
// without bank conflicts).  The three phases of an iteration — rollout, sets, backward sweep — are serial anyway (the
// backward sweep starts at the last knot, the rollout ends there), so one wavefront loses nothing by doing them in turn;
// four tiles per CU sit on four SIMDs and keep their matrix cores busy independently.
//
// Where the state lives: the duals AND the previous slack of every set are REGISTERS of the sets layout (15 + 18 floats
// per knot group and lane for the rocket: 429 at N = 50 — the wavefront is alone on its SIMD and has the 512-entry file to
// itself); LDS holds only the cells.  Nothing of the iterated state reaches HBM during a solve: traffic is x0 in and the
// solution out (+ the workspace in / out when the caller keeps it).
//
// Converged exit with a kept workspace: the reference returns BEFORE `v = vnew` and before the backward sweep, so the
// workspace must hold the slack and the feed-forward term of the iteration BEFORE.  Both are gone by the time an
// instance's convergence is known (the slack registers are updated in place; t_k's cell was reused for u_k).  A lane
// whose own residual terms are all below the tolerances — a necessary condition for its instance to converge at this
// iteration — therefore parks the slack it is about to overwrite directly in the workspace's v / z arrays, and
// d_k = -Kinf x_k - u_k (the rollout's own relation, from the fp32 x_k, u_k of the cells) in its d array; lanes that are
// not locally converged park nothing, so the early iterations cost no traffic.  An instance that leaves at max_iter
// writes its final slack and d = Quu_inv t at exit instead.
//
// Constraint layouts (round 4): the reference takes cone LISTS and linear-inequality row blocks (bindings.cpp:414-490).  In the
// sets layout a second cone of a side is one more in-lane norm over other rows of the same cone slack (its duals behind the
// first cone's), and the rows are a third slack / dual pair of full size per side: the knot's x + dual projected onto one
// half-space after the other (in-lane dot products, coefficients through the scalar cache), exactly the oracle's order.
// Both are compile-time parameters (TransExtra); the built-in entries compile neither, jit.cpp compiles the one kernel of
// a solver's exact layout at its first solve.
#pragma once
#include "admm_mfmac.hip.h"

namespace tmpc {

template <int NX, int NU, int N>
struct TransShape {
    using S = ConeShape<NX, NU>;
    static constexpr int NROW = NX + NU;
    static constexpr int PLEN = 16 * NROW;      // floats of one position: every row x 16 instances
    static constexpr int NG = (N + 3) / 4;      // knot groups of the sets layout
    // Two products per step instead of three.  The stacked vector [x; t] has nx + nu entries, a product takes 4 of them, and
    // the chain of a step is its products one after the other (64 cycles each) + the way back from a result to an operand
    // (~44): three products, 272-280 cycles (measured, also in isolation: experiments/mfma_chain_variants.hip).  Eight
    // entries go through the matrix cores — all of x (slot 0: rows 0..3, slot 1: rows 4..nx-1) and the first MU = min(nu,
    // 8 - nx) input components behind them in slot 1 — the other VU = nu - MU input components and the affine term are
    // multiplied on the VALU into the step's accumulator START, off the chain (they depend on t / r alone).  The
    // accumulator also carries the NEXT step's matrix-core input components through: tile rows nx..7 meet zero operand
    // rows, so what the start value holds there comes out unchanged in the result registers of the lanes that own K slots
    // nx..7 — exactly where the next step's operand is read from.
    static constexpr int NXH = NX > 4 ? NX - 4 : 0;            // state rows in slot 1
    static constexpr int MU = (8 - (4 + NXH)) < NU ? (8 - (4 + NXH)) : NU, VU = NU - MU, VUA = VU > 0 ? VU : 1;
    // coefficient pack, doubles: operand lane fields [NLF][64] (forward slots 0, 1; backward slots 0, 1), the per-lane-group
    // constants [NKC][4] (affine terms, VALU columns), then row-major Pinf [NX][NX], Quu [NU][NU], Quu_inv [NU][NU],
    // Kinf [NU][NX], A [NX][NX], B [NX][NU], f [NX]
    enum { L_MF0 = 0, L_MF1, L_MB0, L_MB1, NLF };
    enum { K_FD0 = 0, K_FD1, K_APF0, K_APF1, K_BPF, K_GF0, K_GF1 = K_GF0 + VUA, K_GF2 = K_GF1 + VUA, K_GB0 = K_GF2 + VUA,
           K_GB1 = K_GB0 + VUA, NKC = K_GB1 + VUA };
    static constexpr int O_KC = NLF * 64, O_PINF = O_KC + NKC * 4, O_QUU = O_PINF + NX * NX, O_QUI = O_QUU + NU * NU,
                         O_KINF = O_QUI + NU * NU, O_A = O_KINF + NU * NX, O_B = O_A + NX * NX, O_F = O_B + NX * NU,
                         COEF_DOUBLES = O_F + NX;
    // (`pi`: per-instance references — a second set of cells, [knot][row][instance], and the terminal term per instance)
    static constexpr size_t lds_floats(int nk, bool pi = false) {
        return (size_t)PLEN * N + ((S::bounds_len(nk) + 1) & ~1) + (pi ? (size_t)PLEN * N : (((size_t)NROW * N + 2) & ~(size_t)1));
    }
    // registers of the sets layout per lane and knot group: duals + previous slack of every set (cxq / cuq: rows of all the
    // cones of a side together; lx / lu: the side has linear-inequality rows — a third slack / dual pair of full size)
    static constexpr int group_regs(int cxq, int cuq, bool lx = false, bool lu = false) {
        return (NX + cxq + NU + cuq) + (NX + (cxq ? NX : 0) + NU + (cuq ? NU : 0)) + (lx ? 2 * NX : 0) + (lu ? 2 * NU : 0);
    }
    static constexpr int state_regs(int cxq, int cuq, bool lx = false, bool lu = false) { return NG * group_regs(cxq, cuq, lx, lu); }
    // the last group(s)' state in LDS instead: when everything together would not fit the 512-entry file — one group where
    // that brings the rest to 470 (config 4: 13 groups of 33, 396 stay), else as many as it takes, three at most (a second
    // state cone or linear rows at N = 50: a tile's LDS grows by 10 KB per group and a CU holds three tiles instead of four)
    static constexpr int spill_groups(int cxq, int cuq, bool lx = false, bool lu = false) {
        if (NG < 2 || state_regs(cxq, cuq, lx, lu) + 70 <= 450) return 0;
        int k = 1;
        while (k < 3 && k < NG - 1 && (NG - k) * group_regs(cxq, cuq, lx, lu) + 70 > 470) ++k;
        return k;
    }
    static constexpr bool spill_last(int cxq, int cuq, bool lx = false, bool lu = false) { return spill_groups(cxq, cuq, lx, lu) > 0; }
    // floats of one such group per lane: its registers, + one entry for each cone array of a side without a cone (the
    // kernel's arrays have at least one entry).  Exactly group_regs for config 4 — whose tile is 592 bytes short of the 40 KB
    // that four tiles per CU allow: six floats more per lane and a CU holds three (4.65 ms instead of 3.14)
    static constexpr int spill_stride(int cxq, int cuq, bool lx = false, bool lu = false) {
        return group_regs(cxq, cuq, lx, lu) + (cxq ? 0 : 2) + (cuq ? 0 : 2);
    }
    static constexpr size_t lds_bytes(int nk, int cxq, int cuq, bool pi = false, bool lx = false, bool lu = false) {
        return sizeof(float) * (lds_floats(nk, pi) + (size_t)64 * spill_groups(cxq, cuq, lx, lu) * spill_stride(cxq, cuq, lx, lu)) +
               sizeof(double) * ((pi ? 16 * NX : 8) + 16 * NX + 4 * NKC);
    }
};

// What the built-in entries do not have and a unit specialised at setup can (jit.cpp): a second cone per side and
// linear-inequality rows (bindings.cpp:414-490 takes cone LISTS and row blocks).  All compile-time, as the first cone is:
// a cone's rows and a row's coefficients are then plain registers / scalar constants of the lane that owns the knot.
struct TransNoExtra {
    static constexpr int CXA2 = 0, CXQ2 = 0, CUA2 = 0, CUQ2 = 0, MLX = 0, MLU = 0;
};
template <int CXA2_, int CXQ2_, int CUA2_, int CUQ2_, int MLX_, int MLU_>
struct TransExtra {
    static constexpr int CXA2 = CXA2_, CXQ2 = CXQ2_, CUA2 = CUA2_, CUQ2 = CUQ2_, MLX = MLX_, MLU = MLU_;
};

// wavefronts per SIMD the register allocation is held to: the state registers + ~80 for everything else
template <int NX, int NU, int N, int CXQ, int CUQ, class GX = TransNoExtra>
constexpr int mfmat_waves_per_simd() {
    // (a second cone, rows and wider states add to the working set of a knot group: held to two wavefronts they spill — 188
    // registers for (8, 2, 12) with two state cones and a row, tests/test_jit.py)
    const int need = TransShape<NX, NU, N>::state_regs(CXQ + GX::CXQ2, CUQ + GX::CUQ2, GX::MLX > 0, GX::MLU > 0) + 80 +
                     4 * (GX::MLX + GX::MLU) + (GX::CXQ2 + GX::CUQ2 > 0 ? 16 : 0) + (NX > 6 ? 8 * (NX - 6) : 0);
    return need <= 128 ? 4 : (need <= 168 ? 3 : (need <= 256 ? 2 : 1));
}

// CXA, CXQ / CUA, CUQ: first row and dimension of the state / input cone (dimension 0: none) — compile-time, so that a
// cone's rows are plain registers of the lane.
template <int NX, int NU, int N, int REFS, int CXA, int CXQ, int CUA, int CUQ, bool BV, class GX = TransNoExtra>
__global__ __launch_bounds__(64, (mfmat_waves_per_simd<NX, NU, N, CXQ, CUQ, GX>())) void admm_mfmat_kernel(const AdmmParams P) {
    using S = ConeShape<NX, NU>;
    using T = TransShape<NX, NU, N>;
    constexpr int XS = S::XS, NROW = S::NROW, PLEN = T::PLEN, NG = T::NG;
    static_assert(N >= 3, "horizon");
    static_assert(NX >= 4, "state slot 0 is full");
    static_assert(CXQ == 0 || (CXQ >= 2 && CXA >= 0 && CXA + CXQ <= NX), "state cone rows");
    static_assert(CUQ == 0 || (CUQ >= 2 && CUA >= 0 && CUA + CUQ <= NU), "input cone rows");
    // second cone of a side (behind the first in the dual arrays; the cone slack is full size anyway), linear rows
    constexpr int CXA2 = GX::CXA2, CXQ2 = GX::CXQ2, CUA2 = GX::CUA2, CUQ2 = GX::CUQ2, MLX = GX::MLX, MLU = GX::MLU;
    static_assert(CXQ2 == 0 || (CXQ > 0 && CXQ2 >= 2 && CXA2 >= CXA + CXQ && CXA2 + CXQ2 <= NX), "second state cone: behind the first, disjoint");
    static_assert(CUQ2 == 0 || (CUQ > 0 && CUQ2 >= 2 && CUA2 >= CUA + CUQ && CUA2 + CUQ2 <= NU), "second input cone: behind the first, disjoint");
    static_assert(MLX >= 0 && MLX <= LIN_MAX_ROWS && MLU >= 0 && MLU <= LIN_MAX_ROWS, "linear rows");
    constexpr bool LX = MLX > 0, LU = MLU > 0;
    constexpr int QX = CXQ + CXQ2, QU = CUQ + CUQ2;            // cone rows of a side
    constexpr int NCX = QX > 0 ? QX : 1, NVX = CXQ > 0 ? NX : 1, NCU = QU > 0 ? QU : 1, NVU = CUQ > 0 ? NU : 1;
    constexpr int NLX = LX ? NX : 1, NLU = LU ? NU : 1;
    extern __shared__ __align__(16) unsigned char s_raw_t[];
    constexpr int nk = BV ? N : 1;   // BV: the bounds depend on the knot (per-knot pack in LDS), else scalars
    float *s_cells = reinterpret_cast<float *>(s_raw_t);
    float *s_bnd = s_cells + (size_t)PLEN * N;
    // shared references: [N][NROW] and one zero cell behind (even offset: fp64 cells follow); per-instance references: a
    // second set of cells [N][NROW][16], staged per tile
    constexpr bool PI = REFS == REF_PER_INSTANCE;
    float *s_ref = s_bnd + ((S::bounds_len(nk) + 1) & ~1);
    double *s_pterm = reinterpret_cast<double *>(s_ref + (PI ? (size_t)PLEN * N : (((size_t)NROW * N + 2) & ~(size_t)1)));   // [NX] (PI: [NX][16])
    double *s_plant = s_pterm + (PI ? 16 * NX : 8);           // closed loop: the plant state of the tile's instances, [16][NX]
    constexpr int KSP = T::spill_groups(QX, QU, LX, LU), GSP = T::spill_stride(QX, QU, LX, LU);   // groups in LDS, floats of one per lane
    constexpr bool SPILL = KSP > 0;
    // every lane's slot addresses stay inside the cells and the reference pack, and a lane without a row only ever reads
    // finite values that meet a zero operand column: the matrix-layout phases then run without lane masks
    constexpr bool FREE = XS == 2 && NROW >= 8 && NU >= 2;
    double *s_kc = s_plant + 16 * NX;                          // per-lane-group constants [NKC][4]
    float *s_last = reinterpret_cast<float *>(s_kc + 4 * T::NKC);   // SPILL: the last knot group's state, [value][64 lanes]

    const int l = threadIdx.x, g = l >> 4, j = l & 15;       // matrix layout: rows g, 4 + g, u row g | sets layout: knot 4 m + g
    const int n_tiles = (P.batch + 15) / 16;
    const long EX = (long)NX * N, EU = (long)NU * (N - 1);
    const int row1 = 4 + g;
    const bool ok1 = XS == 2 && row1 < NX, ok2 = g < NU;

    typedef float __attribute__((address_space(3))) lds_f;
    lds_f *const cm = (lds_f *)s_cells + l;                   // matrix layout: cell (position 0, row g, instance j)
    lds_f *const cq = (lds_f *)s_cells + g * PLEN + j;        // sets layout: cell (position g, row 0, instance j)
    constexpr int U0 = NX * 16;                               // first input row of a position

    // ---- constants ----
    const double *gc64 = reinterpret_cast<const double *>(P.coef);
    // the same pack through the constant address space: wave-uniform reads come through the scalar cache into SGPRs (the
    // matrices of the parking / workspace / plant arithmetic would otherwise take 20-60 VGPRs in branches that rarely run)
    typedef const double __attribute__((address_space(4))) *cdouble_ptr;
    const cdouble_ptr gk64 = (cdouble_ptr)(reinterpret_cast<uintptr_t>(P.coef));
    double cf[T::NLF];
#pragma unroll
    for (int f = 0; f < T::NLF; ++f) cf[f] = gc64[f * 64 + l];
    for (int i = l; i < 4 * T::NKC; i += 64) s_kc[i] = gc64[T::O_KC + i];   // (three or four VALU input columns: more than 64 constants)
    constexpr int NXH = T::NXH, MU = T::MU, VU = T::VU, VUA = T::VUA;
    // this lane's input components: cA (row g of u, what the accumulator's slot 2 starts from / returns), cB (the matrix-core
    // component behind the state rows of slot 1, lanes g >= NXH), and the VU components of the VALU columns; as cell rows
    const bool slot1_x = g < NXH;                             // slot 1 of this lane is a state row (else a passenger)
    const int rowA = NX + (g < NU ? g : NU - 1), rowB = NX + ((g - NXH >= 0 && g - NXH < MU) ? g - NXH : 0);
    auto kc = [&](int field) -> double { return s_kc[field * 4 + g]; };
    auto uni = [](float v) -> float { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const float rho = uni(P.rho), ptol = uni(P.abs_pri_tol), dtol = uni(P.abs_dua_tol);
    float lo_s[NROW], hi_s[NROW];
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
        lo_s[r] = BV ? 0.f : uni(P.bounds[r]);
        hi_s[r] = BV ? 0.f : uni(P.bounds[NROW + r]);
    }
    float mux = 1.f, rmux = 1.f, muu = 1.f, rmuu = 1.f;
    if constexpr (CXQ > 0) mux = uni(P.cx[0]), rmux = uni(1.f / P.cx[0]);
    if constexpr (CUQ > 0) muu = uni(P.cu[0]), rmuu = uni(1.f / P.cu[0]);
    float mux2 = 1.f, rmux2 = 1.f, muu2 = 1.f, rmuu2 = 1.f;
    if constexpr (CXQ2 > 0) mux2 = uni(P.cx[1]), rmux2 = uni(1.f / P.cx[1]);
    if constexpr (CUQ2 > 0) muu2 = uni(P.cu[1]), rmuu2 = uni(1.f / P.cu[1]);
    for (int i = l; i < S::bounds_len(nk); i += 64) s_bnd[i] = P.bounds[i];
    const int ct = P.check_termination;
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;
    const bool keep = P.save_state != 0;                      // the workspace is written back
    const bool park = keep && can_converge && ct > 0;
    const int n_steps = P.mpc_steps > 0 ? P.mpc_steps : 1;
    // The parameter block read afresh where a region needs its pointers: loaded at kernel entry (where the compiler puts
    // kernel-argument loads) the two dozen array pointers of the load / park / store regions stay live across the iteration
    // loop, overflow the SGPR file and come back through v_readlane at every knot group.
    typedef const AdmmParams __attribute__((address_space(4))) *kparam_ptr;
    auto kparams = []() -> kparam_ptr {
        kparam_ptr kp = (kparam_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return kp;
    };

    // shared references of one solve -> LDS: -(Xref .* Q~), -(Uref .* R~) as update_linear_cost forms them (admm.cpp:77-80)
    // per knot, and the terminal term -(Xref_{N-1}' Pinf)' (admm.cpp:81-82)
    auto stage_refs = [&](const float *xref, const float *uref) {
        if constexpr (REFS == REF_SHARED) {
            for (int i = l; i < NROW * N + 1; i += 64) {
                const int k = i / NROW, r = i % NROW;
                float v = 0.f;
                if (i < NROW * N) {
                    if (r < NX) v = -(xref[k * NX + r] * P.bounds[2 * NROW * nk + r]);
                    else if (k < N - 1) v = -(uref[k * NU + (r - NX)] * P.bounds[2 * NROW * nk + r]);
                }
                s_ref[i] = v;
            }
            const double *Pinf = gc64 + T::O_PINF;
            if (l < NX) {
                double acc = 0.0;
                for (int c = 0; c < NX; ++c) acc = fma(Pinf[c * NX + l], (double)xref[(N - 1) * NX + c], acc);   // (Pinf^T xref)[l]
                s_pterm[l] = -acc;
            }
        }
    };
    for (int i = l; i < (PI ? 16 * NX : 8); i += 64) s_pterm[i] = 0.0;
    for (int i = l; i < PLEN * N; i += 64) s_cells[i] = 0.f;   // (cells no lane owns are read by the mask-free phases: keep them finite)
    for (int i = l; i < (PI ? PLEN * N : NROW * N + 2); i += 64) s_ref[i] = 0.f;
    __syncthreads();
    stage_refs(P.xref, P.uref);
    __syncthreads();

    auto cone_scale = [&](float a2, float axv, float mu, float rmu, float &sc, float &ax_new) {
        // The public solver's cone "projection" (restated in oracle/: a <= -mu t -> 0; a <= mu t -> s; else
        // 1/2 (1 + mu t / a) (w, a / mu)): the factor for the head rows and the new axis value
        const float an = __builtin_amdgcn_sqrtf(a2), u0 = axv * mu;
        const bool zero = an <= -u0, keepc = !zero && an <= u0;
        const float half = 0.5f * (1.f + u0 * __builtin_amdgcn_rcpf(an));
        sc = zero ? 0.f : (keepc ? 1.f : half);
        ax_new = zero ? 0.f : (keepc ? axv : half * (an * rmu));
    };
    // squared norm of a cone's head rows, summed in the order the three-wavefront kernels' cross-lane sum has
    // ((h0 + h1) + (h2 + h3), absent rows dropped) and without contraction, so that the two families agree bit for bit
    auto head_norm2 = [&](const float *v, int first, int dim) -> float {
        float h[4] = {0.f, 0.f, 0.f, 0.f};
        bool has[4] = {false, false, false, false};
        float rest = 0.f;
        bool has_rest = false;
#pragma unroll
        for (int c = 0; c < dim - 1; ++c) {
            const int r = first + c;
            const float sq = __fmul_rn(v[r], v[r]);
            if (r < 4) h[r] = sq, has[r] = true;
            else rest = has_rest ? __fadd_rn(rest, sq) : sq, has_rest = true;
        }
        const float p01 = has[0] ? (has[1] ? __fadd_rn(h[0], h[1]) : h[0]) : h[1];
        const float p23 = has[2] ? (has[3] ? __fadd_rn(h[2], h[3]) : h[2]) : h[3];
        float s = (has[0] || has[1]) ? ((has[2] || has[3]) ? __fadd_rn(p01, p23) : p01) : p23;
        if (has_rest) s = (has[0] || has[1] || has[2] || has[3]) ? __fadd_rn(s, rest) : rest;
        return s;
    };

    // ---- the iterated state: registers of the sets layout (group m = knot 4 m + g of instance j) ----
    // Three storage classes, chosen here and not by the register allocator (left to it, the 429 values of config 4 at
    // N = 50 travel in the 3- and 4-register tuples of the wide loads / stores / packed adds that touch them, fragment the
    // file and end up in scratch memory — 100 reloads per iteration, and a lone wavefront sits out every one of them):
    //   arch VGPRs  the duals of the first MD groups (what every iteration reads and writes);
    //   AGPRs       the previous slack of every group — read only by an iteration that forms residuals, written only
    //               where a later one reads it — and the duals of the groups behind MD; one v_accvgpr move per access;
    //   LDS         (SPILL) the last group, where N is not a multiple of 4 and some of its lanes have no knot anyway.
    // Shapes whose state fits the arch VGPRs of their occupancy keep everything there (AREG = false).
    constexpr int NGR = NG - KSP;
    constexpr bool AREG = T::state_regs(QX, QU, LX, LU) + 80 > 256;
    constexpr int RLX = LX ? NX : 0, RLU = LU ? NU : 0;
    constexpr int DG = NX + NCX + NU + NCU + RLX + RLU;         // duals per group
    constexpr int MD = AREG ? (NGR < 150 / DG ? NGR : 150 / DG) : NGR;
    constexpr int MDA = MD > 0 ? MD : 1;
    float a1x[MDA][NX], a2x[MDA][NCX], a1u[MDA][NU], a2u[MDA][NCU];     // duals g, gc | y, yc of the groups in arch VGPRs
    float b1x[NGR][NX], b2x[NGR][NCX], b1u[NGR][NU], b2u[NGR][NCU];     // ... of the groups behind MD (AREG: AGPRs)
    float vbx[NGR][NX], vcx[NGR][NVX], vbu[NGR][NU], vcu[NGR][NVU];     // previous slack v, vc | z, zc (AREG: AGPRs)
    float a3x[MDA][NLX], a3u[MDA][NLU], b3x[NGR][NLX], b3u[NGR][NLU];   // linear rows: duals gl | yl, as a1 / b1
    float vlx[NGR][NLX], vlu[NGR][NLU];                                 // ... and previous slack vl | zl
    auto aget = [](const float &areg) -> float {
        float r;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(areg));
        return r;
    };
    auto aset = [](float &areg, float v) { asm("v_accvgpr_write_b32 %0, %1" : "=a"(areg) : "v"(v)); };
    lds_f *const sl = (lds_f *)s_last + l;
    // value r of array ID of group m, wherever it lives (r a compile-time constant after unrolling)
    enum { S_A1X, S_A2X, S_A1U, S_A2U, S_VBX, S_VCX, S_VBU, S_VCU, S_A3X, S_A3U, S_VLX, S_VLU };
    constexpr int SG_ = DG + NX + NVX + NU + NVU;              // (the four linear-row arrays behind everything else)
    static_assert(SG_ + RLX + RLU == GSP, "a group's LDS image is what TransShape::spill_stride sizes it as");
    auto sld = [&](auto id, auto mt, int r) -> float {
        constexpr int ID = decltype(id)::value, m = decltype(mt)::value;
        constexpr int LOFF[12] = {0, NX, NX + NCX, NX + NCX + NU, DG, DG + NX, DG + NX + NVX, DG + NX + NVX + NU,
                                  NX + NCX + NU + NCU, NX + NCX + NU + NCU + RLX, SG_, SG_ + RLX};
        if constexpr (SPILL && m >= NGR) return sl[64 * ((m - NGR) * GSP + LOFF[ID] + r)];
        else if constexpr (ID == S_A3X) { if constexpr (m < MD) return a3x[m][r]; else return aget(b3x[m][r]); }
        else if constexpr (ID == S_A3U) { if constexpr (m < MD) return a3u[m][r]; else return aget(b3u[m][r]); }
        else if constexpr (ID == S_VLX) { if constexpr (AREG) return aget(vlx[m][r]); else return vlx[m][r]; }
        else if constexpr (ID == S_VLU) { if constexpr (AREG) return aget(vlu[m][r]); else return vlu[m][r]; }
        else if constexpr (ID == S_A1X) { if constexpr (m < MD) return a1x[m][r]; else return aget(b1x[m][r]); }
        else if constexpr (ID == S_A2X) { if constexpr (m < MD) return a2x[m][r]; else return aget(b2x[m][r]); }
        else if constexpr (ID == S_A1U) { if constexpr (m < MD) return a1u[m][r]; else return aget(b1u[m][r]); }
        else if constexpr (ID == S_A2U) { if constexpr (m < MD) return a2u[m][r]; else return aget(b2u[m][r]); }
        else if constexpr (ID == S_VBX) { if constexpr (AREG) return aget(vbx[m][r]); else return vbx[m][r]; }
        else if constexpr (ID == S_VCX) { if constexpr (AREG) return aget(vcx[m][r]); else return vcx[m][r]; }
        else if constexpr (ID == S_VBU) { if constexpr (AREG) return aget(vbu[m][r]); else return vbu[m][r]; }
        else { if constexpr (AREG) return aget(vcu[m][r]); else return vcu[m][r]; }
    };
    auto sst = [&](auto id, auto mt, int r, float v) {
        constexpr int ID = decltype(id)::value, m = decltype(mt)::value;
        constexpr int LOFF[12] = {0, NX, NX + NCX, NX + NCX + NU, DG, DG + NX, DG + NX + NVX, DG + NX + NVX + NU,
                                  NX + NCX + NU + NCU, NX + NCX + NU + NCU + RLX, SG_, SG_ + RLX};
        if constexpr (SPILL && m >= NGR) sl[64 * ((m - NGR) * GSP + LOFF[ID] + r)] = v;
        else if constexpr (ID == S_A3X) { if constexpr (m < MD) a3x[m][r] = v; else aset(b3x[m][r], v); }
        else if constexpr (ID == S_A3U) { if constexpr (m < MD) a3u[m][r] = v; else aset(b3u[m][r], v); }
        else if constexpr (ID == S_VLX) { if constexpr (AREG) aset(vlx[m][r], v); else vlx[m][r] = v; }
        else if constexpr (ID == S_VLU) { if constexpr (AREG) aset(vlu[m][r], v); else vlu[m][r] = v; }
        else if constexpr (ID == S_A1X) { if constexpr (m < MD) a1x[m][r] = v; else aset(b1x[m][r], v); }
        else if constexpr (ID == S_A2X) { if constexpr (m < MD) a2x[m][r] = v; else aset(b2x[m][r], v); }
        else if constexpr (ID == S_A1U) { if constexpr (m < MD) a1u[m][r] = v; else aset(b1u[m][r], v); }
        else if constexpr (ID == S_A2U) { if constexpr (m < MD) a2u[m][r] = v; else aset(b2u[m][r], v); }
        else if constexpr (ID == S_VBX) { if constexpr (AREG) aset(vbx[m][r], v); else vbx[m][r] = v; }
        else if constexpr (ID == S_VCX) { if constexpr (AREG) aset(vcx[m][r], v); else vcx[m][r] = v; }
        else if constexpr (ID == S_VBU) { if constexpr (AREG) aset(vbu[m][r], v); else vbu[m][r] = v; }
        else { if constexpr (AREG) aset(vcu[m][r], v); else vcu[m][r] = v; }
    };
#define LD(ID, r) sld(std::integral_constant<int, ID>{}, mt, r)
#define ST(ID, r, v) sst(std::integral_constant<int, ID>{}, mt, r, v)

    float fm0 = 0.f, fm1 = 0.f, fm2 = 0.f, fm3 = 0.f;          // over this workgroup's tiles: residual maxima, unsolved instances
    int f_unsolved = 0;
    bool refs_shifted = false;                                 // s_ref / s_pterm hold a later step's references than step 0's
    for (;;) {
    // Persistent workgroups: a workgroup takes 16-instance tiles off a global counter until none is left (the hardware
    // dispatcher's strict XCD rotation leaves slots empty, admm_mfmar.hip.h)
    int tk = 0;
    if (l == 0) tk = (int)atomicAdd(&P.gacc[6], 1u);
    const int tile = __builtin_amdgcn_readfirstlane(tk);
    if (tile >= n_tiles) break;
    const long slot_id = (long)tile * 16 + j;
    const bool active = slot_id < P.batch;
    const long b = active ? (P.idx ? (long)P.idx[slot_id] : slot_id) : 0;
    // this lane's knots in the instance-major arrays: element (knot 4 m + g, row r) at ox + 4 m NX + r (ou: the input arrays).
    // 32-bit, and opaque inside the iteration loop: left to itself the compiler forms the 64-bit address of every (array,
    // group) of the parking stores once, outside the loop — 200 registers of addresses that it then spills
    const int ox = (int)(b * EX) + g * NX, ou = (int)(b * EU) + g * NU;

    if constexpr (PI) {
        // per-instance references of the tile -> LDS, as the shared ones: -(Xref .* Q~), -(Uref .* R~) (admm.cpp:77-80) in
        // the cells' own layout [knot][row][instance], and -(Xref_{N-1}' Pinf)' per instance (admm.cpp:81-82).  One flat
        // loop over the tile's 16 x (EX + EU) values: consecutive lanes read consecutive floats of an instance.
        kparam_ptr Pr = kparams();
        const float *xref = Pr->xref, *uref = Pr->uref, *qd = Pr->bounds + 2 * NROW * nk;
        const int bi = (int)b;
        for (int i = l; i < 16 * (int)EX; i += 64) {
            const int jj = i / (int)EX, e = i - jj * (int)EX, k = e / NX, r = e - k * NX;
            const long bj = __shfl(bi, jj, 64);
            s_ref[k * PLEN + r * 16 + jj] = -(xref[bj * EX + e] * qd[r]);
        }
        for (int i = l; i < 16 * (int)EU; i += 64) {
            const int jj = i / (int)EU, e = i - jj * (int)EU, k = e / NU, a = e - k * NU;
            const long bj = __shfl(bi, jj, 64);
            s_ref[k * PLEN + (NX + a) * 16 + jj] = -(uref[bj * EU + e] * qd[NX + a]);
        }
        const double *Pinf = gc64 + T::O_PINF;
        for (int i = l; i < 16 * NX; i += 64) {
            const int row = i >> 4, jj = i & 15;
            const long bj = __shfl(bi, jj, 64);
            double acc = 0.0;
            for (int c = 0; c < NX; ++c) acc = fma(Pinf[c * NX + row], (double)xref[bj * EX + (N - 1) * NX + c], acc);
            s_pterm[row * 16 + jj] = -acc;
        }
        __syncthreads();
    }

    // ---- load the workspace (or the zero workspace tiny_setup leaves, tiny_api.cpp:73-88) ----
    const bool warm = !P.cold_start && active;
    kparam_ptr Pi = kparams();
    mf_for<0, NG>([&](auto mt) {
        float A1x[NX], A2x[NCX], Vbx[NX], Vcx[NVX], A1u[NU], A2u[NCU], Vbu[NU], Vcu[NVU];
        float A3x[NLX], Vlx[NLX], A3u[NLU], Vlu[NLU];
        constexpr int m = decltype(mt)::value;
        const int kk = 4 * m + g;
        const bool xv = warm && kk < N, uv = warm && kk < N - 1;
#pragma unroll
        for (int r = 0; r < NLX; ++r) A3x[r] = 0.f, Vlx[r] = 0.f;
#pragma unroll
        for (int a = 0; a < NLU; ++a) A3u[a] = 0.f, Vlu[a] = 0.f;
        lds_f *c = cq + m * 4 * PLEN;
#pragma unroll
        for (int r = 0; r < NX; ++r) A1x[r] = 0.f, Vbx[r] = 0.f;
#pragma unroll
        for (int r = 0; r < NVX; ++r) Vcx[r] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < NCX; ++c2) A2x[c2] = 0.f;
#pragma unroll
        for (int a = 0; a < NU; ++a) A1u[a] = 0.f, Vbu[a] = 0.f;
#pragma unroll
        for (int a = 0; a < NVU; ++a) Vcu[a] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < NCU; ++c2) A2u[c2] = 0.f;
        if (xv) {
            const float *pg = Pi->sg + ox + m * 4 * NX, *pv = Pi->sv + ox + m * 4 * NX;
#pragma unroll
            for (int r = 0; r < NX; ++r) A1x[r] = pg[r], Vbx[r] = pv[r];
            if constexpr (CXQ > 0) {
                const float *pgc = Pi->sgc + ox + m * 4 * NX, *pvc = Pi->svc + ox + m * 4 * NX;
#pragma unroll
                for (int r = 0; r < NX; ++r) Vcx[r] = pvc[r];
#pragma unroll
                for (int c2 = 0; c2 < CXQ; ++c2) A2x[c2] = pgc[CXA + c2];
#pragma unroll
                for (int c2 = 0; c2 < CXQ2; ++c2) A2x[CXQ + c2] = pgc[CXA2 + c2];
            }
            if constexpr (LX) {
                const float *pgl = Pi->sgl + ox + m * 4 * NX, *pvl = Pi->svl + ox + m * 4 * NX;
#pragma unroll
                for (int r = 0; r < NX; ++r) A3x[r] = pgl[r], Vlx[r] = pvl[r];
            }
        }
        double dv[NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) dv[a] = 0.0;
        if (uv) {
            const float *py = Pi->sy + ou + m * 4 * NU, *pz = Pi->sz + ou + m * 4 * NU, *pd = Pi->sd + ou + m * 4 * NU;
#pragma unroll
            for (int a = 0; a < NU; ++a) A1u[a] = py[a], Vbu[a] = pz[a], dv[a] = (double)pd[a];
            if constexpr (CUQ > 0) {
                const float *pyc = Pi->syc + ou + m * 4 * NU, *pzc = Pi->szc + ou + m * 4 * NU;
#pragma unroll
                for (int a = 0; a < NU; ++a) Vcu[a] = pzc[a];
#pragma unroll
                for (int c2 = 0; c2 < CUQ; ++c2) A2u[c2] = pyc[CUA + c2];
#pragma unroll
                for (int c2 = 0; c2 < CUQ2; ++c2) A2u[CUQ + c2] = pyc[CUA2 + c2];
            }
            if constexpr (LU) {
                const float *pyl = Pi->syl + ou + m * 4 * NU, *pzl = Pi->szl + ou + m * 4 * NU;
#pragma unroll
                for (int a = 0; a < NU; ++a) A3u[a] = pyl[a], Vlu[a] = pzl[a];
            }
        }
        if constexpr (LX) {
#pragma unroll
            for (int r = 0; r < NX; ++r) ST(S_A3X, r, A3x[r]), ST(S_VLX, r, Vlx[r]);
        }
        if constexpr (LU) {
#pragma unroll
            for (int a = 0; a < NU; ++a) ST(S_A3U, a, A3u[a]), ST(S_VLU, a, Vlu[a]);
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) ST(S_A1X, r, A1x[r]), ST(S_VBX, r, Vbx[r]);
#pragma unroll
        for (int r = 0; r < NVX; ++r) ST(S_VCX, r, Vcx[r]);
#pragma unroll
        for (int r = 0; r < NCX; ++r) ST(S_A2X, r, A2x[r]);
#pragma unroll
        for (int a = 0; a < NU; ++a) ST(S_A1U, a, A1u[a]), ST(S_VBU, a, Vbu[a]);
#pragma unroll
        for (int a = 0; a < NVU; ++a) ST(S_VCU, a, Vcu[a]);
#pragma unroll
        for (int a = 0; a < NCU; ++a) ST(S_A2U, a, A2u[a]);
        // the feed-forward term enters as t = Quu d (the rollout's operand carries Quu_inv, admm_mfmac.hip.h)
        if (kk < N - 1) {
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int a2 = 0; a2 < NU; ++a2) acc = fma(gk64[T::O_QUU + a * NU + a2], dv[a2], acc);
                c[U0 + a * 16] = (float)acc;
            }
        }
    });
    double x0r[2];
    if (P.x0d) {
        x0r[0] = active ? P.x0d[b * NX + g] : 0.0;
        x0r[1] = (active && ok1) ? P.x0d[b * NX + row1] : 0.0;
    } else {
        x0r[0] = active ? (double)P.x0[b * NX + g] : 0.0;
        x0r[1] = (active && ok1) ? (double)P.x0[b * NX + row1] : 0.0;
    }
    if (P.mpc_steps > 0) {
        s_plant[j * NX + g] = x0r[0];
        if (ok1) s_plant[j * NX + row1] = x0r[1];
    }

    auto vbu0 = [&](int a) -> float { return sld(std::integral_constant<int, S_VBU>{}, std::integral_constant<int, 0>{}, a); };   // (group 0 is never the one in LDS)
#ifdef TMPC_MFMAT_PROBE
    // timing probe build (scripts/mfmat_cycles.py; the residuals then carry cycle counts, NOT residuals)
    long long T_f = 0, T_s = 0, T_b = 0;
#define TMPC_TPROBE(x) x
#else
#define TMPC_TPROBE(x)
#endif
#ifndef TMPC_MFMAT_HANDOVER
// hand-over stores of the rollout (see `handover`): 0 = plain LDS stores behind a wavefront-scope fence, their order in the
// compiler's assembly checked by tests/test_mfmat_asm.py on every CPU run (default: the compiler pairs and schedules them,
// config 4 3.14 ms); 2 = ONE asm statement per hand-over, an order no compiler can change (3.22 ms: three unpaired stores the
// scheduler cannot place — measured on one box, scripts/ab_variants.py); 1 = 2 without the memory clobber (same time).
// A toolchain on which the assembly test fails builds with -DTMPC_MFMAT_HANDOVER=2 (make MFMAC_FLAGS=-DTMPC_MFMAT_HANDOVER=2).
#define TMPC_MFMAT_HANDOVER 0
#endif
#if TMPC_MFMAT_HANDOVER == 2
#define TMPC_MFMAT_HANDOVER_CLOBBER "memory"
#else
#define TMPC_MFMAT_HANDOVER_CLOBBER
#endif
#ifndef TMPC_MFMAT_SB
#define TMPC_MFMAT_SB 2                                      // steps of the time recurrences per run of matrix-core products
#endif
    int conv = 0, it = 0;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    for (int step = 0; step < n_steps; ++step) {
    if (P.mpc_steps > 0) {
        conv = 0, it = 0;
        if (P.xref_seq && step > 0) {                          // shifted references of this step (rocket_landing_constraints.jl:107-115)
            __syncthreads();
            stage_refs(P.xref_seq + (size_t)step * EX, P.uref_seq + (size_t)step * EU);
            refs_shifted = true;
        } else if (refs_shifted) {                             // a persistent workgroup's next tile: step 0 solves against the
            __syncthreads();                                   // solver's shared references again, not the previous tile's last step
            stage_refs(P.xref, P.uref);
            refs_shifted = false;
        }
    }
    __syncthreads();

    for (int i = 0; i < P.max_iter; ++i) {
        const int itn = i + 1;
        const bool check = ct > 0 && itn % ct == 0;
        const bool need_res = check && (can_converge || itn == last_check_it);
        const bool last = itn == P.max_iter;
        // the previous-slack registers take this iteration's slack where something reads it: the next iteration's
        // residuals, or the solution (an instance can only leave at a check or at max_iter)
        const bool check_next = ct > 0 && (itn + 1) % ct == 0 && itn < P.max_iter;
        const bool upd_old = need_res || last || (check_next && (can_converge || itn + 1 == last_check_it));
        TMPC_TPROBE(const long long tp0 = clock64();)
        // ================= rollout (admm.cpp:25-35), matrix layout =================
        // x+ = (A - B Kinf) x - B Quu_inv t + f,  u = -Kinf x - Quu_inv t  with t = B'p + r kept by the backward sweep.
        // Step k:  c_k = [MF0 MF1] [x_k; t_k(matrix-core components)] + start_k,  start_k = f + (VALU columns) t_k(rest) with
        // the passengers of step k + 1 in the slot-1 rows no state row owns.  The two products are issued back to back the
        // moment the previous step's result registers — which ARE their operands — are there; then, while the matrix core
        // runs them: the previous step's results are converted and handed to the sets (x_k, u_{k-1}), and the next step's
        // start is formed (from cells read two steps ago) in the tuple they came from.  Two accumulator tuples swap roles
        // from step to step; nothing is copied and no vector instruction stands between a result and the products that need it.
        // Stores without lane masks (FREE): a lane without a row in slot 1 / 2 lands on its position's input cells / on the
        // next position's first state cells; both are written — by the lanes that own them — AFTER it.
        {
            constexpr int STEPS = N - 1;
            lds_f *pp = cm;                                    // matrix-layout cell (position k, row g, instance j)
            lds_f *const pu = (lds_f *)s_cells + j;            // + position * PLEN + row * 16
            auto ldV = [&](int pos, float (&tv)[VUA]) {
#pragma unroll
                for (int v = 0; v < VU; ++v) tv[v] = pu[pos * PLEN + (NX + MU + v) * 16];
            };
            auto ldB = [&](int pos) -> float { return pu[pos * PLEN + rowB * 16]; };
            const double k_fd0 = kc(T::K_FD0), k_fd1 = kc(T::K_FD1);
            double k_gf0[VUA], k_gf1[VUA], k_gf2[VUA];
#pragma unroll
            for (int v = 0; v < VUA; ++v) k_gf0[v] = kc(T::K_GF0 + v), k_gf1[v] = kc(T::K_GF1 + v), k_gf2[v] = kc(T::K_GF2 + v);
            // the accumulator start of a step, written into the tuple in place (slot 3 — tile rows 12.. — stays the zero it is)
            auto start = [&](mf_d4 &c, const double (&tvd)[VUA], double tb_next) {
                double a0 = k_fd0, a1 = k_fd1, a2 = 0.0;
#pragma unroll
                for (int v = 0; v < VU; ++v) a0 = fma(k_gf0[v], tvd[v], a0), a1 = fma(k_gf1[v], tvd[v], a1), a2 = fma(k_gf2[v], tvd[v], a2);
                if (!slot1_x) a1 = tb_next;                    // passenger: the next step's matrix-core component of t
                c[0] = a0, c[1] = a1, c[2] = a2;
            };
            auto handover = [&](const mf_d4 &c, auto kc) {      // x_k (k = 0: x0 itself — the cell held the sets' sum), u_{k-1}
                constexpr int k = decltype(kc)::value;
                if constexpr (FREE && TMPC_MFMAT_HANDOVER == 0) {   // plain stores, fence, the assembly test (tests/test_mfmat_asm.py)
                    // (the order of a mask-free store and the owner's store to the same cell is an order between LANES, which the
                    // compiler does not see — to it the two addresses never alias, and it has moved one across the other; LDS
                    // executes a wavefront's stores in program order, so a wavefront-scope fence — no instruction — pins it)
                    lds_f *pk = pp + k * PLEN;
                    if (k > 0) pk[U0 - PLEN] = (float)c[2];
                    pk[64] = (float)c[1];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    pk[0] = (float)c[0];
                } else if constexpr (FREE) {
                    // The order of a mask-free store and the owner's store to the same cell is an order between LANES, which the
                    // compiler does not see — to it the two addresses never alias, and it has moved one across the other (knots 25
                    // and 37 of every instance came out wrong).  LDS executes a wavefront's stores in program order, so the stores
                    // of a hand-over are ONE asm statement (fixed order inside it; volatile asm statements keep their order among
                    // themselves; the memory clobber keeps the compiler's own LDS accesses on their side of it): the form for a
                    // toolchain on which the default's assembly test fails.
                    const unsigned col = (unsigned)(size_t)pp;                  // LDS byte address of this lane's column of cells
                    const float f0 = (float)c[0], f1 = (float)c[1];
                    if constexpr (k > 0) {
                        const float f2 = (float)c[2];                           // (t_{k-1} is spent)
                        asm volatile("ds_write_b32 %0, %1 offset:%4\n\tds_write_b32 %0, %2 offset:%5\n\tds_write_b32 %0, %3 offset:%6"
                                     :
                                     : "v"(col), "v"(f2), "v"(f1), "v"(f0), "n"(4 * ((k - 1) * PLEN + U0)), "n"(4 * (k * PLEN + 64)),
                                       "n"(4 * k * PLEN)
                                     : TMPC_MFMAT_HANDOVER_CLOBBER);
                    } else {
                        asm volatile("ds_write_b32 %0, %1 offset:%3\n\tds_write_b32 %0, %2 offset:%4"
                                     :
                                     : "v"(col), "v"(f1), "v"(f0), "n"(4 * (k * PLEN + 64)), "n"(4 * k * PLEN)
                                     : TMPC_MFMAT_HANDOVER_CLOBBER);
                    }
                } else {
                    lds_f *pk = pp + k * PLEN;
                    if (k > 0 && ok2) pk[U0 - PLEN] = (float)c[2];
                    if (ok1) pk[64] = (float)c[1];
                    pk[0] = (float)c[0];
                }
            };
            // (order of the work, from experiments/mfmat_cost_probe.hip: an fp64 product holds the wavefront's issue for ~66 cycles
            // whatever follows, and the FIRST vector / LDS instruction behind products costs a flat ~55 cycles more, the
            // rest ~4 each — so products go back to back and the vector work in one block behind them; split around the
            // second product of a step it pays the flat cost twice: 281 against 255 cycles per step.  And the flat cost is per
            // block, not per step: SB steps' products — step k + 1's operands ARE step k's result registers — are issued as
            // one run, SB + 1 accumulator tuples taking turns, and the hand-overs and next starts of all SB steps follow as one block.)
            constexpr int SB = TMPC_MFMAT_SB, NT = SB + 1;
            mf_d4 c[NT];
            float tvn[SB][VUA], tbn[SB];
            auto ld = [&](int k, int sl) {                      // what the start of step k is formed from (cells the sweep left)
                if (k < STEPS) ldV(k, tvn[sl]);
                tbn[sl] = k + 1 < STEPS ? ldB(k + 1) : 0.f;
            };
            auto start_of = [&](mf_d4 &cc, int sl) {
                double tvd[VUA];
#pragma unroll
                for (int v = 0; v < VUA; ++v) tvd[v] = (double)tvn[sl][v];
                start(cc, tvd, (double)tbn[sl]);
            };
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2) c[t2] = mf_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sl = 0; sl < SB; ++sl)
#pragma unroll
                for (int v = 0; v < VUA; ++v) tvn[sl][v] = 0.f;
            c[0][0] = x0r[0], c[0][1] = slot1_x ? x0r[1] : (double)ldB(0);
            mf_for<0, SB>([&](auto st) {
                constexpr int sl = decltype(st)::value;
                if constexpr (sl < STEPS) {
                    ld(sl, sl);
                    start_of(c[sl + 1], sl);
                }
            });
            mf_for<0, SB>([&](auto st) {
                constexpr int sl = decltype(st)::value;
                if constexpr (SB + sl < STEPS) ld(SB + sl, sl);
            });
            mf_for<0, (STEPS + SB - 1) / SB>([&](auto bt) {
                constexpr int k0 = decltype(bt)::value * SB;
                constexpr int nb = STEPS - k0 < SB ? STEPS - k0 : SB;
                mf_for<0, nb>([&](auto st) {
                    constexpr int k = k0 + decltype(st)::value;
                    c[(k + 1) % NT] = mf_mma(cf[T::L_MF0], c[k % NT][0], c[(k + 1) % NT]);
                    c[(k + 1) % NT] = mf_mma(cf[T::L_MF1], c[k % NT][1], c[(k + 1) % NT]);
                });
                __builtin_amdgcn_sched_barrier(0);             // (left to itself the scheduler puts a hand-over in front of the products)
                mf_for<0, nb>([&](auto st) {
                    constexpr int k = k0 + decltype(st)::value;
                    handover(c[k % NT], std::integral_constant<int, k>{});
                });
                mf_for<0, SB>([&](auto st) {
                    constexpr int sl = decltype(st)::value, k = k0 + SB + sl;  // a step of the next run: its tuple held x_{k - SB}, handed over above
                    if constexpr (k < STEPS) {
                        start_of(c[(k + 1) % NT], sl);
                        if constexpr (k + SB < STEPS) ld(k + SB, sl);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            handover(c[STEPS % NT], std::integral_constant<int, STEPS>{});
            // (slot 3 of the tuples — tile rows 12.. — is never read; "used" here so that the register allocator does not
            // park temporaries in it: a vector write into a tuple that a product in flight is about to overwrite waits for it)
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2) asm volatile("" ::"v"(c[t2][3]));
        }
        __syncthreads();
        TMPC_TPROBE(const long long tp1 = clock64(); T_f += tp1 - tp0;)
        // ================= the sets (admm.cpp:43-69) and the residual terms (admm.cpp:93-96), sets layout =================
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
        int oxi = ox, oui = ou;
        asm volatile("" : "+v"(oxi), "+v"(oui));
        if (!conv) {                                           // a converged instance's state is frozen
            // a group's cells are read while the group before it is worked on (a lone wavefront sits out every LDS latency it
            // has not hidden itself); its reference terms are needed only at the END of its arithmetic and are read at its
            // start — a group ahead they would be nine more registers across the group with the most live values.  Lanes
            // past the last knot read inside the allocation, unused
            // linear-inequality rows (specialised units): coefficients, right-hand sides and |a|^2 through the scalar cache,
            // read here and not at kernel entry — they are scalar registers of the sets phase only
            float lax[MLX > 0 ? MLX : 1][NX], lbx[MLX > 0 ? MLX : 1], lnx[MLX > 0 ? MLX : 1];
            float lau[MLU > 0 ? MLU : 1][NU], lbu[MLU > 0 ? MLU : 1], lnu[MLU > 0 ? MLU : 1];
            if constexpr (LX || LU) {
                typedef const float __attribute__((address_space(4))) *cfloat_ptr;
                const cfloat_ptr lp = (cfloat_ptr)(reinterpret_cast<uintptr_t>(kparams()->lin));
                constexpr int OU = MLX * (NX + 2);             // (the pack holds the enabled sides: solver.hip prepare_launch)
#pragma unroll
                for (int k = 0; k < MLX; ++k) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) lax[k][r] = lp[k * NX + r];
                    lbx[k] = lp[MLX * NX + k], lnx[k] = lp[MLX * NX + MLX + k];
                }
#pragma unroll
                for (int k = 0; k < MLU; ++k) {
#pragma unroll
                    for (int a = 0; a < NU; ++a) lau[k][a] = lp[OU + k * NU + a];
                    lbu[k] = lp[OU + MLU * NU + k], lnu[k] = lp[OU + MLU * NU + MLU + k];
                }
            }
            float nx_[NX], nu_[NU];
            auto fetch = [&](int m) {
                lds_f *c = cq + m * 4 * PLEN;
#pragma unroll
                for (int r = 0; r < NX; ++r) nx_[r] = c[r * 16];
#pragma unroll
                for (int a = 0; a < NU; ++a) nu_[a] = c[U0 + a * 16];
            };
            fetch(0);
            mf_for<0, NG>([&](auto mt) {
                __builtin_amdgcn_sched_barrier(0);             // one group at a time: interleaved, their temporaries overflow the file
                constexpr int m = decltype(mt)::value;
                constexpr bool x_all = 4 * m + 3 < N, u_all = 4 * m + 3 < N - 1;
                const int kk = 4 * m + g;
                lds_f *c = cq + m * 4 * PLEN;
                const bool xv = x_all || kk < N, uv = u_all || kk < N - 1;   // (compile-time true except in the last group(s))
                float x[NX], u[NU], rfx[NX], rfu[NU];
#pragma unroll
                for (int r = 0; r < NX; ++r) x[r] = nx_[r], rfx[r] = REFS == REF_SHARED ? s_ref[kk * NROW + r] : 0.f;
#pragma unroll
                for (int a = 0; a < NU; ++a) u[a] = nu_[a], rfu[a] = REFS == REF_SHARED ? s_ref[kk * NROW + NX + a] : 0.f;   // (the cell behind the last knot's is zero)
                if constexpr (PI) {                            // this instance's own: the reference cells mirror the cells (the last knot's input rows stay zero)
                    const lds_f *rc = (const lds_f *)s_ref + (x_all ? kk : (kk < N ? kk : N - 1)) * PLEN + j;
#pragma unroll
                    for (int r = 0; r < NX; ++r) rfx[r] = rc[r * 16];
#pragma unroll
                    for (int a = 0; a < NU; ++a) rfu[a] = rc[U0 + a * 16];
                }
                if constexpr (m + 1 < NG) fetch(m + 1);
                // (the terminal knot's reference enters through Pinf, admm.cpp:81-82, not through q)
                if constexpr (REFS != REF_ZERO && 4 * m + 3 >= N - 1) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) rfx[r] = kk == N - 1 ? 0.f : rfx[r];
                }
                // ---- all the arithmetic of the group first, in ONE basic block: the box and cone sets of the state rows and
                // of the input rows are four independent chains (a cone's sqrt -> rcp -> selects chain alone is ~120 cycles of
                // dependent issue for a lone wavefront), which the scheduler can only interleave if no branch separates them.
                // Lanes past the last knot compute on whatever the allocation holds; only their stores and their residual
                // terms are masked.
                float sx[NX], vn[NX], vc[NVX], su[NU], zn[NU], zc[NVU];
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    const float lo = BV ? s_bnd[(xv ? kk : 0) * 2 * NROW + r] : lo_s[r], hi = BV ? s_bnd[(xv ? kk : 0) * 2 * NROW + NROW + r] : hi_s[r];
                    const float w = x[r] + LD(S_A1X, r);
                    vn[r] = __builtin_amdgcn_fmed3f(w, lo, hi);                      // admm.cpp:52-56
                    const float an = w - vn[r];                                      // admm.cpp:68
                    ST(S_A1X, r, an);
                    sx[r] = vn[r] - an;
                }
#pragma unroll
                for (int a = 0; a < NU; ++a) {
                    const float lo = BV ? s_bnd[(uv ? kk : 0) * 2 * NROW + NX + a] : lo_s[NX + a],
                                hi = BV ? s_bnd[(uv ? kk : 0) * 2 * NROW + NROW + NX + a] : hi_s[NX + a];
                    const float w = u[a] + LD(S_A1U, a);
                    zn[a] = __builtin_amdgcn_fmed3f(w, lo, hi);
                    const float an = w - zn[a];
                    ST(S_A1U, a, an);
                    su[a] = zn[a] - an;
                }
                if constexpr (CXQ > 0) {
                    // every state row carries the cone set's slack and dual (the solver's arrays are full size); for a row
                    // outside the cone the "projection" is the identity: slack = x, the dual stays zero
                    float w2[NCX], a2n[NCX];
#pragma unroll
                    for (int r = 0; r < NX; ++r) vc[r] = x[r];
#pragma unroll
                    for (int c2 = 0; c2 < CXQ; ++c2) w2[c2] = x[CXA + c2] + LD(S_A2X, c2), vc[CXA + c2] = w2[c2];
#pragma unroll
                    for (int c2 = 0; c2 < CXQ2; ++c2) w2[CXQ + c2] = x[CXA2 + c2] + LD(S_A2X, CXQ + c2), vc[CXA2 + c2] = w2[CXQ + c2];
                    float sc, ax_new;
                    cone_scale(head_norm2(vc, CXA, CXQ), vc[CXA + CXQ - 1], mux, rmux, sc, ax_new);
#pragma unroll
                    for (int c2 = 0; c2 < CXQ - 1; ++c2) vc[CXA + c2] *= sc;
                    vc[CXA + CXQ - 1] = ax_new;
                    if constexpr (CXQ2 > 0) {
                        cone_scale(head_norm2(vc, CXA2, CXQ2), vc[CXA2 + CXQ2 - 1], mux2, rmux2, sc, ax_new);
#pragma unroll
                        for (int c2 = 0; c2 < CXQ2 - 1; ++c2) vc[CXA2 + c2] *= sc;
                        vc[CXA2 + CXQ2 - 1] = ax_new;
                    }
#pragma unroll
                    for (int c2 = 0; c2 < CXQ; ++c2) a2n[c2] = w2[c2] - vc[CXA + c2], ST(S_A2X, c2, a2n[c2]);
#pragma unroll
                    for (int c2 = 0; c2 < CXQ2; ++c2) a2n[CXQ + c2] = w2[CXQ + c2] - vc[CXA2 + c2], ST(S_A2X, CXQ + c2, a2n[CXQ + c2]);
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        const bool in1 = r >= CXA && r < CXA + CXQ, in2 = CXQ2 > 0 && r >= CXA2 && r < CXA2 + CXQ2;
                        sx[r] += in1 ? vc[r] - a2n[in1 ? r - CXA : 0] : (in2 ? vc[r] - a2n[in2 ? CXQ + r - CXA2 : 0] : x[r]);
                    }
                }
                float vl[NLX], zl[NLU];
                if constexpr (LX) {
                    // third set of the state rows: the knot's x + dual projected onto one half-space after the other
                    // (bindings.cpp:414-450; restated in oracle/: project_halfspaces), all in this lane
                    float w3[NX];
#pragma unroll
                    for (int r = 0; r < NX; ++r) w3[r] = x[r] + LD(S_A3X, r), vl[r] = w3[r];
#pragma unroll
                    for (int k = 0; k < MLX; ++k) {
                        float dot = 0.f;
#pragma unroll
                        for (int r = 0; r < NX; ++r) dot = fmaf(lax[k][r], vl[r], dot);
                        const float t = (dot > lbx[k] && lnx[k] > 0.f) ? (dot - lbx[k]) / lnx[k] : 0.f;
#pragma unroll
                        for (int r = 0; r < NX; ++r) vl[r] -= t * lax[k][r];
                    }
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        const float an = w3[r] - vl[r];
                        ST(S_A3X, r, an);
                        sx[r] += vl[r] - an;
                    }
                }
                if constexpr (CUQ > 0) {
                    float w2[NCU], a2n[NCU];
#pragma unroll
                    for (int a = 0; a < NU; ++a) zc[a] = u[a];
#pragma unroll
                    for (int c2 = 0; c2 < CUQ; ++c2) w2[c2] = u[CUA + c2] + LD(S_A2U, c2), zc[CUA + c2] = w2[c2];
#pragma unroll
                    for (int c2 = 0; c2 < CUQ2; ++c2) w2[CUQ + c2] = u[CUA2 + c2] + LD(S_A2U, CUQ + c2), zc[CUA2 + c2] = w2[CUQ + c2];
                    float sc, ax_new;
                    cone_scale(head_norm2(zc, CUA, CUQ), zc[CUA + CUQ - 1], muu, rmuu, sc, ax_new);
#pragma unroll
                    for (int c2 = 0; c2 < CUQ - 1; ++c2) zc[CUA + c2] *= sc;
                    zc[CUA + CUQ - 1] = ax_new;
                    if constexpr (CUQ2 > 0) {
                        cone_scale(head_norm2(zc, CUA2, CUQ2), zc[CUA2 + CUQ2 - 1], muu2, rmuu2, sc, ax_new);
#pragma unroll
                        for (int c2 = 0; c2 < CUQ2 - 1; ++c2) zc[CUA2 + c2] *= sc;
                        zc[CUA2 + CUQ2 - 1] = ax_new;
                    }
#pragma unroll
                    for (int c2 = 0; c2 < CUQ; ++c2) a2n[c2] = w2[c2] - zc[CUA + c2], ST(S_A2U, c2, a2n[c2]);
#pragma unroll
                    for (int c2 = 0; c2 < CUQ2; ++c2) a2n[CUQ + c2] = w2[CUQ + c2] - zc[CUA2 + c2], ST(S_A2U, CUQ + c2, a2n[CUQ + c2]);
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        const bool in1 = a >= CUA && a < CUA + CUQ, in2 = CUQ2 > 0 && a >= CUA2 && a < CUA2 + CUQ2;
                        su[a] += in1 ? zc[a] - a2n[in1 ? a - CUA : 0] : (in2 ? zc[a] - a2n[in2 ? CUQ + a - CUA2 : 0] : u[a]);
                    }
                }
                if constexpr (LU) {
                    float w3[NU];
#pragma unroll
                    for (int a = 0; a < NU; ++a) w3[a] = u[a] + LD(S_A3U, a), zl[a] = w3[a];
#pragma unroll
                    for (int k = 0; k < MLU; ++k) {
                        float dot = 0.f;
#pragma unroll
                        for (int a = 0; a < NU; ++a) dot = fmaf(lau[k][a], zl[a], dot);
                        const float t = (dot > lbu[k] && lnu[k] > 0.f) ? (dot - lbu[k]) / lnu[k] : 0.f;
#pragma unroll
                        for (int a = 0; a < NU; ++a) zl[a] -= t * lau[k][a];
                    }
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        const float an = w3[a] - zl[a];
                        ST(S_A3U, a, an);
                        su[a] += zl[a] - an;
                    }
                }
                // what the backward sweep needs of a row is the linear-cost term (admm.cpp:77-80), not the sum itself:
                // q = -(Xref Q~) - rho (sum over sets of slack - dual), r likewise
                if (xv) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) c[r * 16] = rfx[r] - rho * sx[r];
                }
                if (uv) {
#pragma unroll
                    for (int a = 0; a < NU; ++a) c[U0 + a * 16] = rfu[a] - rho * su[a];
                }
                // ---- residual terms (admm.cpp:93-96) and, where this iteration may be the instance's last, the parking ----
                if (need_res) {
                    float gp = 0.f, gd = 0.f, hp = 0.f, hd = 0.f;
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        gp = fmaxf(gp, fabsf(x[r] - vn[r]));
                        gd = fmaxf(gd, fabsf(LD(S_VBX, r) - vn[r]));
                    }
                    if constexpr (CXQ > 0) {
#pragma unroll
                        for (int r = 0; r < NX; ++r) {
                            if ((r >= CXA && r < CXA + CXQ) || (CXQ2 > 0 && r >= CXA2 && r < CXA2 + CXQ2)) gp = fmaxf(gp, fabsf(x[r] - vc[r]));
                            gd = fmaxf(gd, fabsf(LD(S_VCX, r) - vc[r]));
                        }
                    }
                    if constexpr (LX) {
#pragma unroll
                        for (int r = 0; r < NX; ++r) {
                            gp = fmaxf(gp, fabsf(x[r] - vl[r]));
                            gd = fmaxf(gd, fabsf(LD(S_VLX, r) - vl[r]));
                        }
                    }
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        hp = fmaxf(hp, fabsf(u[a] - zn[a]));
                        hd = fmaxf(hd, fabsf(LD(S_VBU, a) - zn[a]));
                    }
                    if constexpr (CUQ > 0) {
#pragma unroll
                        for (int a = 0; a < NU; ++a) {
                            if ((a >= CUA && a < CUA + CUQ) || (CUQ2 > 0 && a >= CUA2 && a < CUA2 + CUQ2)) hp = fmaxf(hp, fabsf(u[a] - zc[a]));
                            hd = fmaxf(hd, fabsf(LD(S_VCU, a) - zc[a]));
                        }
                    }
                    if constexpr (LU) {
#pragma unroll
                        for (int a = 0; a < NU; ++a) {
                            hp = fmaxf(hp, fabsf(u[a] - zl[a]));
                            hd = fmaxf(hd, fabsf(LD(S_VLU, a) - zl[a]));
                        }
                    }
                    if (xv) pri_x = fmaxf(pri_x, gp), dua_x = fmaxf(dua_x, gd);
                    if (uv) pri_u = fmaxf(pri_u, hp), dua_u = fmaxf(dua_u, hd);
                    if (park && active && xv && gp < ptol && gd * rho < dtol) {          // this iteration may be the instance's last
                        kparam_ptr Pk = kparams();
                        float *pv = Pk->sv + oxi + m * 4 * NX;
#pragma unroll
                        for (int r = 0; r < NX; ++r) pv[r] = LD(S_VBX, r);
                        if constexpr (CXQ > 0) {
                            float *pvc = Pk->svc + oxi + m * 4 * NX;
#pragma unroll
                            for (int r = 0; r < NX; ++r) pvc[r] = LD(S_VCX, r);
                        }
                        if constexpr (LX) {
                            float *pvl = Pk->svl + oxi + m * 4 * NX;
#pragma unroll
                            for (int r = 0; r < NX; ++r) pvl[r] = LD(S_VLX, r);
                        }
                    }
                    if (park && active && uv && hp < ptol && hd * rho < dtol) {
                        kparam_ptr Pk = kparams();
                        float *pz = Pk->sz + oui + m * 4 * NU, *pd = Pk->sd + oui + m * 4 * NU;
#pragma unroll
                        for (int a = 0; a < NU; ++a) pz[a] = LD(S_VBU, a);
                        if constexpr (CUQ > 0) {
                            float *pzc = Pk->szc + oui + m * 4 * NU;
#pragma unroll
                            for (int a = 0; a < NU; ++a) pzc[a] = LD(S_VCU, a);
                        }
                        if constexpr (LU) {
                            float *pzl = Pk->szl + oui + m * 4 * NU;
#pragma unroll
                            for (int a = 0; a < NU; ++a) pzl[a] = LD(S_VLU, a);
                        }
                        // the feed-forward term this iteration's rollout used: d = -Kinf x - u (admm.cpp:29)
#pragma unroll
                        for (int a = 0; a < NU; ++a) {
                            double acc = 0.0;
#pragma unroll
                            for (int r = 0; r < NX; ++r) acc = fma(gk64[T::O_KINF + a * NX + r], (double)x[r], acc);
                            pd[a] = (float)(-acc - (double)u[a]);
                        }
                    }
                }
                // ---- the previous-slack registers take this iteration's slack where something will read it ----
                if (upd_old) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        ST(S_VBX, r, vn[r]);
                        if constexpr (CXQ > 0) ST(S_VCX, r, vc[r]);
                        if constexpr (LX) ST(S_VLX, r, vl[r]);
                    }
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        ST(S_VBU, a, zn[a]);
                        if constexpr (CUQ > 0) ST(S_VCU, a, zc[a]);
                        if constexpr (LU) ST(S_VLU, a, zl[a]);
                    }
                }
            });
        }
        __syncthreads();
        TMPC_TPROBE(const long long tp2 = clock64(); T_s += tp2 - tp1;)
        // ================= termination (admm.cpp:89-107, :181-193), per instance =================
        if (!conv) it = itn;
        if (need_res) {
            const float r0 = mf_inst_max(pri_x), r1 = mf_inst_max(dua_x) * rho, r2 = mf_inst_max(pri_u), r3 = mf_inst_max(dua_u) * rho;
            if (!conv) {
                res0 = r0, res1 = r1, res2 = r2, res3 = r3;
                if (res0 < ptol && res2 < ptol && res1 < dtol && res3 < dtol) conv = 1;
            }
        }
        const bool any_left = __builtin_amdgcn_ballot_w64(active && !conv) != 0ull;
        if (!any_left || (last && !keep)) break;               // (a one-shot solve has no use for the last backward sweep)
        // ================= fused backward sweep (admm.cpp:75-83, :13-20), matrix layout =================
        // stage kn: p_kn = q_kn + AmBKt p_{kn+1} - Kinf' r_kn (+ AmBKt Pinf f),  t_kn = B' p_{kn+1} + r_kn (+ B' Pinf f) with
        // q_kn = -(Xref Q~) - rho s_x(kn), r_kn = -(Uref R~) - rho s_u(kn).  As in the rollout: two products per stage on
        // [p_{kn+1}; r_kn(matrix-core components)], the rest of r_kn and the constants in the accumulator's start, the
        // passengers r_{kn-1} riding in it; the start of stage kn - 1 is formed under the products of stage kn from cells
        // read a stage earlier, t_{kn+1} goes to its cell there too.
        {
            lds_f *const pc = (lds_f *)s_cells + j;            // + position * PLEN + row * 16
            const int row1c = slot1_x ? row1 : g;              // (a passenger lane re-reads its slot-0 row: the value is not used)
            auto lin = [&](int kn, int row) -> float { return pc[kn * PLEN + row * 16]; };   // q / r of a row at a knot (the sets left it)
            struct Ops {
                float q0, q1, ra, rb, rv[VUA];
            };
            auto operands = [&](int kn) -> Ops {               // of the start of stage kn (passenger: r_{kn-1})
                Ops o;
                o.q0 = lin(kn, g), o.q1 = lin(kn, row1c), o.ra = lin(kn, rowA);
                o.rb = kn >= 1 ? lin(kn - 1, rowB) : 0.f;
#pragma unroll
                for (int v = 0; v < VUA; ++v) o.rv[v] = v < VU ? lin(kn, NX + MU + v) : 0.f;
                return o;
            };
            const double k_apf0 = kc(T::K_APF0), k_apf1 = kc(T::K_APF1), k_bpf = kc(T::K_BPF);
            double k_gb0[VUA], k_gb1[VUA];
#pragma unroll
            for (int v = 0; v < VUA; ++v) k_gb0[v] = kc(T::K_GB0 + v), k_gb1[v] = kc(T::K_GB1 + v);
            struct OpsD {
                double q0, q1, ra, rb, rv[VUA];
            };
            auto widen = [&](const Ops &o) -> OpsD {
                OpsD d;
                d.q0 = (double)o.q0, d.q1 = (double)o.q1, d.ra = (double)o.ra, d.rb = (double)o.rb;
#pragma unroll
                for (int v = 0; v < VUA; ++v) d.rv[v] = (double)o.rv[v];
                return d;
            };
            auto start = [&](mf_d4 &c, const OpsD &o) {
                double a0 = o.q0 + k_apf0, a1 = o.q1 + k_apf1, a2 = o.ra + k_bpf;
#pragma unroll
                for (int v = 0; v < VU; ++v) a0 = fma(k_gb0[v], o.rv[v], a0), a1 = fma(k_gb1[v], o.rv[v], a1);
                if (!slot1_x) a1 = o.rb;                       // passenger: the next stage's matrix-core component of r
                c[0] = a0, c[1] = a1, c[2] = a2;
            };
            // stage index s = N - 2 - kn ascending; stage s reads the tuple s % NT (p_{kn+1}, t_{kn+1}) and accumulates into the
            // next one; runs of SB stages as in the rollout
            constexpr int STG = N - 1, SB = TMPC_MFMAT_SB, NT = SB + 1;
            mf_d4 c[NT];
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2) c[t2] = mf_d4{0.0, 0.0, 0.0, 0.0};
            {
                double pt0 = 0.0, pt1 = 0.0;
                if constexpr (REFS == REF_SHARED) pt0 = s_pterm[g], pt1 = s_pterm[row1c];
                if constexpr (PI) pt0 = s_pterm[g * 16 + j], pt1 = s_pterm[row1c * 16 + j];
                const double p0 = pt0 + (double)pc[(N - 1) * PLEN + g * 16];                   // admm.cpp:81-82
                const double p1 = pt1 + (double)pc[(N - 1) * PLEN + row1c * 16];
                c[0][0] = p0, c[0][1] = slot1_x ? p1 : (double)lin(N - 2, rowB);
            }
            Ops on[SB];
            mf_for<0, SB>([&](auto st) {
                constexpr int sl = decltype(st)::value;
                if constexpr (sl < STG) start(c[sl + 1], widen(operands(N - 2 - sl)));
            });
            mf_for<0, SB>([&](auto st) {
                constexpr int sl = decltype(st)::value;
                on[sl] = operands(SB + sl < STG ? N - 2 - SB - sl : 0);
            });
            lds_f *const tw = cm + U0;                                               // + position * PLEN: the cell of a knot's t
            const bool tmask = ok2 && !conv;                                         // (a converged instance keeps its feed-forward term)
            mf_for<0, (STG + SB - 1) / SB>([&](auto bt) {
                constexpr int s0 = decltype(bt)::value * SB;
                constexpr int nb = STG - s0 < SB ? STG - s0 : SB;
                mf_for<0, nb>([&](auto st) {
                    constexpr int sg = s0 + decltype(st)::value;
                    c[(sg + 1) % NT] = mf_mma(cf[T::L_MB0], c[sg % NT][0], c[(sg + 1) % NT]);   // [AmBKt; B'] p (+ -Kinf' r)
                    c[(sg + 1) % NT] = mf_mma(cf[T::L_MB1], c[sg % NT][1], c[(sg + 1) % NT]);
                });
                __builtin_amdgcn_sched_barrier(0);
                mf_for<0, nb>([&](auto st) {
                    constexpr int sg = s0 + decltype(st)::value, kn = N - 2 - sg;
                    if (kn < N - 2 && tmask) tw[(kn + 1) * PLEN] = (float)c[sg % NT][2];        // t_{kn+1}
                });
                mf_for<0, SB>([&](auto st) {
                    constexpr int sl = decltype(st)::value, sg = s0 + SB + sl;
                    if constexpr (sg < STG) {
                        start(c[(sg + 1) % NT], widen(on[sl]));
                        if constexpr (sg + SB < STG) on[sl] = operands(N - 2 - sg - SB);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            if (tmask) tw[0] = (float)c[STG % NT][2];                                // t_0
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2) asm volatile("" ::"v"(c[t2][3]));
        }
        __syncthreads();
        TMPC_TPROBE(T_b += clock64() - tp2;)
        if (last) break;
    }
    TMPC_TPROBE(res0 = (float)T_f / (float)P.max_iter; res1 = (float)T_s / (float)P.max_iter; res2 = (float)T_b / (float)P.max_iter; res3 = 0.f;)

    // ================= results of the solve =================
    if (P.mpc_steps > 0) {
        // closed loop (cartpole_example_mpc.jl:35-51, rocket_landing_constraints.jl:119-123): apply the first control to the
        // model in fp64, log, and go on from the new state with the workspace as it stands
        kparam_ptr Pe = kparams();
        if (g == 0) {
            const long so = b * P.mpc_steps + step;
            double xn[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                double acc = gk64[T::O_F + r];
#pragma unroll
                for (int c = 0; c < NX; ++c) acc = fma(gk64[T::O_A + r * NX + c], s_plant[j * NX + c], acc);
#pragma unroll
                for (int a = 0; a < NU; ++a) acc = fma(gk64[T::O_B + r * NU + a], (double)vbu0(a), acc);
                xn[r] = acc;
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                s_plant[j * NX + r] = xn[r];
                if (active) Pe->mpc_x[so * NX + r] = (float)xn[r];
            }
            if (active) {
#pragma unroll
                for (int a = 0; a < NU; ++a) Pe->mpc_u[so * NU + a] = vbu0(a);
                Pe->mpc_iter[so] = conv ? it : -it;
            }
        }
        __syncthreads();
        x0r[0] = s_plant[j * NX + g];
        x0r[1] = ok1 ? s_plant[j * NX + row1] : 0.0;
        if (step + 1 < n_steps) {
            // An instance that converged goes into its next solve exactly as the reference does: with the slack and the
            // feed-forward term of the iteration BEFORE its last (it parked them in the workspace arrays, see above) — the
            // registers hold the last iteration's slack, and its t cells were overwritten by that iteration's rollout.
            if (conv) {
                mf_for<0, NG>([&](auto mt) {
                    constexpr int m = decltype(mt)::value;
                    const int kk = 4 * m + g;
                    lds_f *c = cq + m * 4 * PLEN;
                    if (kk < N) {
                        const float *pv = Pe->sv + ox + m * 4 * NX;
#pragma unroll
                        for (int r = 0; r < NX; ++r) ST(S_VBX, r, pv[r]);
                        if constexpr (CXQ > 0) {
                            const float *pvc = Pe->svc + ox + m * 4 * NX;
#pragma unroll
                            for (int r = 0; r < NX; ++r) ST(S_VCX, r, pvc[r]);
                        }
                        if constexpr (LX) {
                            const float *pvl = Pe->svl + ox + m * 4 * NX;
#pragma unroll
                            for (int r = 0; r < NX; ++r) ST(S_VLX, r, pvl[r]);
                        }
                    }
                    if (kk < N - 1) {
                        const float *pz = Pe->sz + ou + m * 4 * NU, *pd = Pe->sd + ou + m * 4 * NU;
#pragma unroll
                        for (int a = 0; a < NU; ++a) ST(S_VBU, a, pz[a]);
                        if constexpr (CUQ > 0) {
                            const float *pzc = Pe->szc + ou + m * 4 * NU;
#pragma unroll
                            for (int a = 0; a < NU; ++a) ST(S_VCU, a, pzc[a]);
                        }
                        if constexpr (LU) {
                            const float *pzl = Pe->szl + ou + m * 4 * NU;
#pragma unroll
                            for (int a = 0; a < NU; ++a) ST(S_VLU, a, pzl[a]);
                        }
                        double dv[NU];
#pragma unroll
                        for (int a = 0; a < NU; ++a) dv[a] = (double)pd[a];
#pragma unroll
                        for (int a = 0; a < NU; ++a) {
                            double acc = 0.0;
#pragma unroll
                            for (int a2 = 0; a2 < NU; ++a2) acc = fma(gk64[T::O_QUU + a * NU + a2], dv[a2], acc);
                            c[U0 + a * 16] = (float)acc;
                        }
                    }
                });
            }
            __syncthreads();
            continue;
        }
        if (active) {
            Pe->x0_out[b * NX + g] = (float)x0r[0];
            if (ok1) Pe->x0_out[b * NX + row1] = (float)x0r[1];
        }
    }
    }   // closed-loop steps

    // the projected slack of the last executed iteration is the solution (admm.cpp:187-188,204-205)
    kparam_ptr Pe = kparams();
    mf_for<0, NG>([&](auto mt) {
        float A1x[NX], A2x[NCX], Vbx[NX], Vcx[NVX], A1u[NU], A2u[NCU], Vbu[NU], Vcu[NVU];
#pragma unroll
        for (int r = 0; r < NX; ++r) A1x[r] = LD(S_A1X, r), Vbx[r] = LD(S_VBX, r);
#pragma unroll
        for (int r = 0; r < NVX; ++r) Vcx[r] = LD(S_VCX, r);
#pragma unroll
        for (int r = 0; r < NCX; ++r) A2x[r] = LD(S_A2X, r);
#pragma unroll
        for (int a = 0; a < NU; ++a) A1u[a] = LD(S_A1U, a), Vbu[a] = LD(S_VBU, a);
#pragma unroll
        for (int a = 0; a < NVU; ++a) Vcu[a] = LD(S_VCU, a);
#pragma unroll
        for (int a = 0; a < NCU; ++a) A2u[a] = LD(S_A2U, a);
        float A3x[NLX], Vlx[NLX], A3u[NLU], Vlu[NLU];
#pragma unroll
        for (int r = 0; r < NLX; ++r) A3x[r] = LX ? LD(S_A3X, r) : 0.f, Vlx[r] = LX ? LD(S_VLX, r) : 0.f;
#pragma unroll
        for (int a = 0; a < NLU; ++a) A3u[a] = LU ? LD(S_A3U, a) : 0.f, Vlu[a] = LU ? LD(S_VLU, a) : 0.f;
        constexpr int m = decltype(mt)::value;
        const int kk = 4 * m + g;
        lds_f *c = cq + m * 4 * PLEN;
        if (active && kk < N) {
            float *po = Pe->xout + ox + m * 4 * NX;
#pragma unroll
            for (int r = 0; r < NX; ++r) po[r] = Vbx[r];
            if (keep) {
                float *pg = Pe->sg + ox + m * 4 * NX;
#pragma unroll
                for (int r = 0; r < NX; ++r) pg[r] = A1x[r];
                if constexpr (CXQ > 0) {
                    float *pgc = Pe->sgc + ox + m * 4 * NX;
#pragma unroll
                    for (int c2 = 0; c2 < CXQ; ++c2) pgc[CXA + c2] = A2x[c2];
#pragma unroll
                    for (int c2 = 0; c2 < CXQ2; ++c2) pgc[CXA2 + c2] = A2x[CXQ + c2];
                }
                if constexpr (LX) {
                    float *pgl = Pe->sgl + ox + m * 4 * NX;
#pragma unroll
                    for (int r = 0; r < NX; ++r) pgl[r] = A3x[r];
                }
                if (!conv) {                                   // (a converged instance parked the slack of the iteration before)
                    float *pv = Pe->sv + ox + m * 4 * NX;
#pragma unroll
                    for (int r = 0; r < NX; ++r) pv[r] = Vbx[r];
                    if constexpr (CXQ > 0) {
                        float *pvc = Pe->svc + ox + m * 4 * NX;
#pragma unroll
                        for (int r = 0; r < NX; ++r) pvc[r] = Vcx[r];
                    }
                    if constexpr (LX) {
                        float *pvl = Pe->svl + ox + m * 4 * NX;
#pragma unroll
                        for (int r = 0; r < NX; ++r) pvl[r] = Vlx[r];
                    }
                }
            }
        }
        if (active && kk < N - 1) {
            float *po = Pe->uout + ou + m * 4 * NU;
#pragma unroll
            for (int a = 0; a < NU; ++a) po[a] = Vbu[a];
            if (keep) {
                float *py = Pe->sy + ou + m * 4 * NU;
#pragma unroll
                for (int a = 0; a < NU; ++a) py[a] = A1u[a];
                if constexpr (CUQ > 0) {
                    float *pyc = Pe->syc + ou + m * 4 * NU;
#pragma unroll
                    for (int c2 = 0; c2 < CUQ; ++c2) pyc[CUA + c2] = A2u[c2];
#pragma unroll
                    for (int c2 = 0; c2 < CUQ2; ++c2) pyc[CUA2 + c2] = A2u[CUQ + c2];
                }
                if constexpr (LU) {
                    float *pyl = Pe->syl + ou + m * 4 * NU;
#pragma unroll
                    for (int a = 0; a < NU; ++a) pyl[a] = A3u[a];
                }
                if (!conv) {
                    float *pz = Pe->sz + ou + m * 4 * NU, *pd = Pe->sd + ou + m * 4 * NU;
#pragma unroll
                    for (int a = 0; a < NU; ++a) pz[a] = Vbu[a];
                    if constexpr (CUQ > 0) {
                        float *pzc = Pe->szc + ou + m * 4 * NU;
#pragma unroll
                        for (int a = 0; a < NU; ++a) pzc[a] = Vcu[a];
                    }
                    if constexpr (LU) {
                        float *pzl = Pe->szl + ou + m * 4 * NU;
#pragma unroll
                        for (int a = 0; a < NU; ++a) pzl[a] = Vlu[a];
                    }
                    double tv[NU];                             // d = Quu_inv t of the last backward sweep (admm.cpp:17)
#pragma unroll
                    for (int a = 0; a < NU; ++a) tv[a] = (double)c[U0 + a * 16];
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        double acc = 0.0;
#pragma unroll
                        for (int a2 = 0; a2 < NU; ++a2) acc = fma(gk64[T::O_QUI + a * NU + a2], tv[a2], acc);
                        pd[a] = (float)acc;
                    }
                }
            }
        }
    });
    if (active && g == 0) {
        Pe->iter[b] = Pe->iter_offset + it;
        Pe->solved[b] = conv;
        Pe->res[b * 4 + 0] = res0;
        Pe->res[b * 4 + 1] = res1;
        Pe->res[b * 4 + 2] = res2;
        Pe->res[b * 4 + 3] = res3;
    }
    fm0 = fmaxf(fm0, active ? res0 : 0.f), fm1 = fmaxf(fm1, active ? res1 : 0.f);
    fm2 = fmaxf(fm2, active ? res2 : 0.f), fm3 = fmaxf(fm3, active ? res3 : 0.f);
    f_unsolved += __popcll(__builtin_amdgcn_ballot_w64(active && !conv && g == 0));
    __syncthreads();                                           // the tile's cells are free for the next one
    }   // tile loop
    {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            fm0 = fmaxf(fm0, __shfl_xor(fm0, off, 64));
            fm1 = fmaxf(fm1, __shfl_xor(fm1, off, 64));
            fm2 = fmaxf(fm2, __shfl_xor(fm2, off, 64));
            fm3 = fmaxf(fm3, __shfl_xor(fm3, off, 64));
        }
        fold_status(P, fm0, fm1, fm2, fm3, f_unsolved, l);
    }
}

#undef LD
#undef ST
#undef TMPC_TPROBE
}  // namespace tmpc
