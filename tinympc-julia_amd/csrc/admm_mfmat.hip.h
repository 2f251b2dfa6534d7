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
#pragma once
#include "admm_mfmac.hip.h"

namespace tmpc {

template <int NX, int NU, int N>
struct TransShape {
    using S = ConeShape<NX, NU>;
    static constexpr int NROW = NX + NU;
    static constexpr int PLEN = 16 * NROW;      // floats of one position: every row x 16 instances
    static constexpr int NG = (N + 3) / 4;      // knot groups of the sets layout
    // coefficient pack, doubles: ConeShape's lane fields [NF][64], then row-major Pinf [NX][NX] (as mfmac), Quu [NU][NU],
    // Quu_inv [NU][NU], Kinf [NU][NX], A [NX][NX], B [NX][NU], f [NX]
    static constexpr int O_PINF = S::NF * 64, O_QUU = O_PINF + NX * NX, O_QUI = O_QUU + NU * NU, O_KINF = O_QUI + NU * NU,
                         O_A = O_KINF + NU * NX, O_B = O_A + NX * NX, O_F = O_B + NX * NU, COEF_DOUBLES = O_F + NX;
    static constexpr size_t lds_floats(int nk) {
        return (size_t)PLEN * N + ((S::bounds_len(nk) + 1) & ~1) + (((size_t)NROW * N + 2) & ~(size_t)1);
    }
    static constexpr size_t lds_bytes(int nk) { return sizeof(float) * lds_floats(nk) + sizeof(double) * (8 + 16 * NX); }
    // registers of the sets layout per lane: duals + previous slack of every set
    static constexpr int state_regs(int cxq, int cuq) {
        return NG * ((NX + cxq + NU + cuq) + (NX + (cxq ? NX : 0) + NU + (cuq ? NU : 0)));
    }
};

// wavefronts per SIMD the register allocation is held to: the state registers + ~80 for everything else
template <int NX, int NU, int N, int CXQ, int CUQ>
constexpr int mfmat_waves_per_simd() {
    const int need = TransShape<NX, NU, N>::state_regs(CXQ, CUQ) + 80;
    return need <= 128 ? 4 : (need <= 168 ? 3 : (need <= 256 ? 2 : 1));
}

// CXA, CXQ / CUA, CUQ: first row and dimension of the state / input cone (dimension 0: none) — compile-time, so that a
// cone's rows are plain registers of the lane.
template <int NX, int NU, int N, int REFS, int CXA, int CXQ, int CUA, int CUQ, bool BV>
__global__ __launch_bounds__(64, (mfmat_waves_per_simd<NX, NU, N, CXQ, CUQ>())) void admm_mfmat_kernel(const AdmmParams P) {
    using S = ConeShape<NX, NU>;
    using T = TransShape<NX, NU, N>;
    constexpr int XS = S::XS, NROW = S::NROW, PLEN = T::PLEN, NG = T::NG;
    static_assert(N >= 3, "horizon");
    static_assert(NX >= 4, "state slot 0 is full");
    static_assert(CXQ == 0 || (CXQ >= 2 && CXA >= 0 && CXA + CXQ <= NX), "state cone rows");
    static_assert(CUQ == 0 || (CUQ >= 2 && CUA >= 0 && CUA + CUQ <= NU), "input cone rows");
    constexpr int NCX = CXQ > 0 ? CXQ : 1, NVX = CXQ > 0 ? NX : 1, NCU = CUQ > 0 ? CUQ : 1, NVU = CUQ > 0 ? NU : 1;
    extern __shared__ __align__(16) unsigned char s_raw_t[];
    constexpr int nk = BV ? N : 1;   // BV: the bounds depend on the knot (per-knot pack in LDS), else scalars
    float *s_cells = reinterpret_cast<float *>(s_raw_t);
    float *s_bnd = s_cells + (size_t)PLEN * N;
    float *s_ref = s_bnd + ((S::bounds_len(nk) + 1) & ~1);    // [N][NROW] and one zero cell behind (even offset: fp64 cells follow)
    double *s_pterm = reinterpret_cast<double *>(s_ref + (((size_t)NROW * N + 2) & ~(size_t)1));
    double *s_plant = s_pterm + 8;                            // closed loop: the plant state of the tile's instances, [16][NX]

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
    double cf[S::NF];
#pragma unroll
    for (int f = 0; f < S::NF; ++f) cf[f] = gc64[f * 64 + l];
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
    for (int i = l; i < S::bounds_len(nk); i += 64) s_bnd[i] = P.bounds[i];
    const int ct = P.check_termination;
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;
    const bool keep = P.save_state != 0;                      // the workspace is written back
    const bool park = keep && can_converge && ct > 0;
    constexpr bool ONE_COL = NU <= 3;                         // the affine term rides in the products (admm_mfmac.hip.h)
    const bool one_lane = ONE_COL && g == 3;
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
    float a1x[NG][NX], a2x[NG][NCX], vbx[NG][NX], vcx[NG][NVX];     // state rows: box dual g, cone dual gc | previous slack v, vc
    float a1u[NG][NU], a2u[NG][NCU], vbu[NG][NU], vcu[NG][NVU];     // input rows: y, yc | z, zc

    float fm0 = 0.f, fm1 = 0.f, fm2 = 0.f, fm3 = 0.f;          // over this workgroup's tiles: residual maxima, unsolved instances
    int f_unsolved = 0;
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

    // ---- load the workspace (or the zero workspace tiny_setup leaves, tiny_api.cpp:73-88) ----
    const bool warm = !P.cold_start && active;
    kparam_ptr Pi = kparams();
    mf_for<0, NG>([&](auto mt) {
        constexpr int m = decltype(mt)::value;
        const int kk = 4 * m + g;
        const bool xv = warm && kk < N, uv = warm && kk < N - 1;
        lds_f *c = cq + m * 4 * PLEN;
#pragma unroll
        for (int r = 0; r < NX; ++r) a1x[m][r] = 0.f, vbx[m][r] = 0.f;
#pragma unroll
        for (int r = 0; r < NVX; ++r) vcx[m][r] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < NCX; ++c2) a2x[m][c2] = 0.f;
#pragma unroll
        for (int a = 0; a < NU; ++a) a1u[m][a] = 0.f, vbu[m][a] = 0.f;
#pragma unroll
        for (int a = 0; a < NVU; ++a) vcu[m][a] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < NCU; ++c2) a2u[m][c2] = 0.f;
        if (xv) {
            const float *pg = Pi->sg + ox + m * 4 * NX, *pv = Pi->sv + ox + m * 4 * NX;
#pragma unroll
            for (int r = 0; r < NX; ++r) a1x[m][r] = pg[r], vbx[m][r] = pv[r];
            if constexpr (CXQ > 0) {
                const float *pgc = Pi->sgc + ox + m * 4 * NX, *pvc = Pi->svc + ox + m * 4 * NX;
#pragma unroll
                for (int r = 0; r < NX; ++r) vcx[m][r] = pvc[r];
#pragma unroll
                for (int c2 = 0; c2 < CXQ; ++c2) a2x[m][c2] = pgc[CXA + c2];
            }
        }
        double dv[NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) dv[a] = 0.0;
        if (uv) {
            const float *py = Pi->sy + ou + m * 4 * NU, *pz = Pi->sz + ou + m * 4 * NU, *pd = Pi->sd + ou + m * 4 * NU;
#pragma unroll
            for (int a = 0; a < NU; ++a) a1u[m][a] = py[a], vbu[m][a] = pz[a], dv[a] = (double)pd[a];
            if constexpr (CUQ > 0) {
                const float *pyc = Pi->syc + ou + m * 4 * NU, *pzc = Pi->szc + ou + m * 4 * NU;
#pragma unroll
                for (int a = 0; a < NU; ++a) vcu[m][a] = pzc[a];
#pragma unroll
                for (int c2 = 0; c2 < CUQ; ++c2) a2u[m][c2] = pyc[CUA + c2];
            }
        }
        // the feed-forward term enters as t = Quu d (the rollout's operand carries Quu_inv, admm_mfmac.hip.h)
        if (kk < N - 1) {
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int a2 = 0; a2 < NU; ++a2) acc = fma(gc64[T::O_QUU + a * NU + a2], dv[a2], acc);
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

    int conv = 0, it = 0;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    for (int step = 0; step < n_steps; ++step) {
    if (P.mpc_steps > 0) {
        conv = 0, it = 0;
        if (P.xref_seq && step > 0) {                          // shifted references of this step (rocket_landing_constraints.jl:107-115)
            __syncthreads();
            stage_refs(P.xref_seq + (size_t)step * EX, P.uref_seq + (size_t)step * EU);
        }
    }
    __syncthreads();

    for (int i = 0; i < P.max_iter; ++i) {
        const int itn = i + 1;
        const bool check = ct > 0 && itn % ct == 0;
        const bool need_res = check && (can_converge || itn == last_check_it);
        const bool last = itn == P.max_iter;
        // ================= rollout (admm.cpp:25-35), matrix layout =================
        // x+ = (A - B Kinf) x - B Quu_inv t + f,  u = -Kinf x - Quu_inv t  with t = B'p + r kept by the backward sweep;
        // the product with t does not depend on x: the one of step k + 1 is issued behind the x products of step k
        {
            auto t_product = [&](float t) -> mf_d4 {
                mf_d4 c = {0.0, 0.0, 0.0, 0.0};
                if constexpr (!ONE_COL) c[0] = cf[S::F_FD0], c[1] = cf[S::F_FD1];
                return mf_mma(cf[S::F_MF2], (double)(one_lane ? 1.f : t), c);
            };
            lds_f *pp = cm;                                    // position k
            pp[0] = (float)x0r[0];                             // knot 0 for the sets (x0 is given; the cell held the sets' sum)
            if (ok1) pp[64] = (float)x0r[1];
            float t0 = 0.f, t_next = 0.f;
            if (ok2) t0 = pp[U0], t_next = pp[PLEN + U0];
            mf_d4 cpre = t_product(t0);
            double xa = x0r[0], xb = x0r[1];
            for (int k = 0; k < N - 1; ++k) {
                mf_d4 c = mf_mma(cf[S::F_MF0], xa, cpre);
                if constexpr (XS == 2) c = mf_mma(cf[S::F_MF1], xb, c);
                if (k + 1 < N - 1) {
                    cpre = t_product(t_next);
                    if (k + 2 < N - 1 && ok2) t_next = pp[2 * PLEN + U0];
                }
                xa = c[0], xb = c[1];
                pp[PLEN] = (float)xa;                          // x_{k+1}
                if (ok1) pp[PLEN + 64] = (float)xb;
                if (ok2) pp[U0] = (float)c[2];                 // u_k (t_k is spent)
                pp += PLEN;
            }
        }
        __syncthreads();
        // ================= the sets (admm.cpp:43-69) and the residual terms (admm.cpp:93-96), sets layout =================
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
        int oxi = ox, oui = ou;
        asm volatile("" : "+v"(oxi), "+v"(oui));
        if (!conv) {                                           // a converged instance's state is frozen
            mf_for<0, NG>([&](auto mt) {
                constexpr int m = decltype(mt)::value;
                constexpr bool x_all = 4 * m + 3 < N, u_all = 4 * m + 3 < N - 1, u_any = 4 * m < N - 1;
                const int kk = 4 * m + g;
                lds_f *c = cq + m * 4 * PLEN;
                if (x_all || kk < N) {
                    float x[NX], sx[NX], vn[NX], vc[NVX];
#pragma unroll
                    for (int r = 0; r < NX; ++r) x[r] = c[r * 16];
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        const float lo = BV ? s_bnd[kk * 2 * NROW + r] : lo_s[r], hi = BV ? s_bnd[kk * 2 * NROW + NROW + r] : hi_s[r];
                        const float w = x[r] + a1x[m][r];
                        vn[r] = __builtin_amdgcn_fmed3f(w, lo, hi);                  // admm.cpp:52-56
                        a1x[m][r] = w - vn[r];                                       // admm.cpp:68
                        sx[r] = vn[r] - a1x[m][r];
                    }
                    if constexpr (CXQ > 0) {
                        // every state row carries the cone set's slack and dual (the solver's arrays are full size); for a row
                        // outside the cone the "projection" is the identity: slack = x, the dual stays zero
                        float w2[NCX];
#pragma unroll
                        for (int r = 0; r < NX; ++r) vc[r] = x[r];
#pragma unroll
                        for (int c2 = 0; c2 < CXQ; ++c2) w2[c2] = x[CXA + c2] + a2x[m][c2], vc[CXA + c2] = w2[c2];
                        float sc, ax_new;
                        cone_scale(head_norm2(vc, CXA, CXQ), vc[CXA + CXQ - 1], mux, rmux, sc, ax_new);
#pragma unroll
                        for (int c2 = 0; c2 < CXQ - 1; ++c2) vc[CXA + c2] *= sc;
                        vc[CXA + CXQ - 1] = ax_new;
#pragma unroll
                        for (int c2 = 0; c2 < CXQ; ++c2) a2x[m][c2] = w2[c2] - vc[CXA + c2];
#pragma unroll
                        for (int r = 0; r < NX; ++r) {
                            const bool in = r >= CXA && r < CXA + CXQ;
                            sx[r] += in ? vc[r] - a2x[m][in ? r - CXA : 0] : x[r];
                        }
                    }
                    if (need_res) {
                        float gp = 0.f, gd = 0.f;
#pragma unroll
                        for (int r = 0; r < NX; ++r) {
                            gp = fmaxf(gp, fabsf(x[r] - vn[r]));
                            gd = fmaxf(gd, fabsf(vbx[m][r] - vn[r]));
                        }
                        if constexpr (CXQ > 0) {
#pragma unroll
                            for (int r = 0; r < NX; ++r) {
                                if (r >= CXA && r < CXA + CXQ) gp = fmaxf(gp, fabsf(x[r] - vc[r]));
                                gd = fmaxf(gd, fabsf(vcx[m][r] - vc[r]));
                            }
                        }
                        pri_x = fmaxf(pri_x, gp), dua_x = fmaxf(dua_x, gd);
                        if (park && active && gp < ptol && gd * rho < dtol) {          // this iteration may be the instance's last
                            kparam_ptr Pk = kparams();
                            float *pv = Pk->sv + oxi + m * 4 * NX;
#pragma unroll
                            for (int r = 0; r < NX; ++r) pv[r] = vbx[m][r];
                            if constexpr (CXQ > 0) {
                                float *pvc = Pk->svc + oxi + m * 4 * NX;
#pragma unroll
                                for (int r = 0; r < NX; ++r) pvc[r] = vcx[m][r];
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        vbx[m][r] = vn[r];
                        if constexpr (CXQ > 0) vcx[m][r] = vc[r];
                        c[r * 16] = sx[r];
                    }
                    if constexpr (u_any) {
                        if (u_all || kk < N - 1) {
                            float u[NU], su[NU], zn[NU], zc[NVU];
#pragma unroll
                            for (int a = 0; a < NU; ++a) u[a] = c[U0 + a * 16];
#pragma unroll
                            for (int a = 0; a < NU; ++a) {
                                const float lo = BV ? s_bnd[kk * 2 * NROW + NX + a] : lo_s[NX + a],
                                            hi = BV ? s_bnd[kk * 2 * NROW + NROW + NX + a] : hi_s[NX + a];
                                const float w = u[a] + a1u[m][a];
                                zn[a] = __builtin_amdgcn_fmed3f(w, lo, hi);
                                a1u[m][a] = w - zn[a];
                                su[a] = zn[a] - a1u[m][a];
                            }
                            if constexpr (CUQ > 0) {
                                float w2[NCU];
#pragma unroll
                                for (int a = 0; a < NU; ++a) zc[a] = u[a];
#pragma unroll
                                for (int c2 = 0; c2 < CUQ; ++c2) w2[c2] = u[CUA + c2] + a2u[m][c2], zc[CUA + c2] = w2[c2];
                                float sc, ax_new;
                                cone_scale(head_norm2(zc, CUA, CUQ), zc[CUA + CUQ - 1], muu, rmuu, sc, ax_new);
#pragma unroll
                                for (int c2 = 0; c2 < CUQ - 1; ++c2) zc[CUA + c2] *= sc;
                                zc[CUA + CUQ - 1] = ax_new;
#pragma unroll
                                for (int c2 = 0; c2 < CUQ; ++c2) a2u[m][c2] = w2[c2] - zc[CUA + c2];
#pragma unroll
                                for (int a = 0; a < NU; ++a) {
                                    const bool in = a >= CUA && a < CUA + CUQ;
                                    su[a] += in ? zc[a] - a2u[m][in ? a - CUA : 0] : u[a];
                                }
                            }
                            if (need_res) {
                                float gp = 0.f, gd = 0.f;
#pragma unroll
                                for (int a = 0; a < NU; ++a) {
                                    gp = fmaxf(gp, fabsf(u[a] - zn[a]));
                                    gd = fmaxf(gd, fabsf(vbu[m][a] - zn[a]));
                                }
                                if constexpr (CUQ > 0) {
#pragma unroll
                                    for (int a = 0; a < NU; ++a) {
                                        if (a >= CUA && a < CUA + CUQ) gp = fmaxf(gp, fabsf(u[a] - zc[a]));
                                        gd = fmaxf(gd, fabsf(vcu[m][a] - zc[a]));
                                    }
                                }
                                pri_u = fmaxf(pri_u, gp), dua_u = fmaxf(dua_u, gd);
                                if (park && active && gp < ptol && gd * rho < dtol) {
                                    kparam_ptr Pk = kparams();
                                    float *pz = Pk->sz + oui + m * 4 * NU, *pd = Pk->sd + oui + m * 4 * NU;
#pragma unroll
                                    for (int a = 0; a < NU; ++a) pz[a] = vbu[m][a];
                                    if constexpr (CUQ > 0) {
                                        float *pzc = Pk->szc + oui + m * 4 * NU;
#pragma unroll
                                        for (int a = 0; a < NU; ++a) pzc[a] = vcu[m][a];
                                    }
                                    // the feed-forward term this iteration's rollout used: d = -Kinf x - u (admm.cpp:29)
#pragma unroll
                                    for (int a = 0; a < NU; ++a) {
                                        double acc = 0.0;
#pragma unroll
                                        for (int r = 0; r < NX; ++r) acc = fma(gc64[T::O_KINF + a * NX + r], (double)x[r], acc);
                                        pd[a] = (float)(-acc - (double)u[a]);
                                    }
                                }
                            }
#pragma unroll
                            for (int a = 0; a < NU; ++a) {
                                vbu[m][a] = zn[a];
                                if constexpr (CUQ > 0) vcu[m][a] = zc[a];
                                c[U0 + a * 16] = su[a];
                            }
                        }
                    }
                }
            });
        }
        __syncthreads();
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
        // q_kn = -(Xref Q~) - rho s_x(kn), r_kn = -(Uref R~) - rho s_u(kn); only the products with p are on the recurrence: a
        // stage's accumulator start {q, r} and its product with r are formed and issued one stage ahead, behind the previous
        // stage's p products, from operands read two stages ahead
        {
            lds_f *qp = cm + (N - 1) * PLEN;                   // position of the operands read next
            double p[2];
            {
                double pt0 = 0.0, pt1 = 0.0;
                if constexpr (REFS == REF_SHARED) {
                    pt0 = s_pterm[g];
                    pt1 = ok1 ? s_pterm[row1] : 0.0;
                }
                float s0 = qp[0], s1 = 0.f;
                if (ok1) s1 = qp[64];
                p[0] = pt0 - (double)(rho * s0);                                     // admm.cpp:81-82
                p[1] = pt1 - (double)(rho * s1);
            }
            auto stage_operands = [&](int kn, float (&sv)[3], float (&rf)[3]) {      // s and reference terms of knot kn
                qp -= PLEN;
                sv[0] = qp[0];
                sv[1] = 0.f, sv[2] = 0.f;
                if (ok1) sv[1] = qp[64];
                if (ok2) sv[2] = qp[U0];
                rf[0] = rf[1] = rf[2] = 0.f;
                if constexpr (REFS == REF_SHARED) {
                    rf[0] = s_ref[kn * NROW + g];
                    if (ok1) rf[1] = s_ref[kn * NROW + row1];
                    if (ok2) rf[2] = s_ref[kn * NROW + NX + g];
                }
            };
            auto stage_start = [&](int kn, const float (&sv)[3], const float (&rf)[3]) -> mf_d4 {
                const double r_in = (double)(rf[2] - rho * sv[2]);                   // admm.cpp:77-78
                mf_d4 c = {(double)(rf[0] - rho * sv[0]), (double)(rf[1] - rho * sv[1]), r_in, 0.0};   // admm.cpp:79-80
                if (kn == 0) c[0] = c[1] = 0.0;                                      // q_0 enters p_0 only, which nothing reads
                if constexpr (!ONE_COL) c[0] += cf[S::F_APF0], c[1] += cf[S::F_APF1], c[2] += cf[S::F_BPF];
                return mf_mma(cf[S::F_MB2], one_lane ? 1.0 : r_in, c);               // [-Kinf^T; 0] r
            };
            float sv[3], rf[3];
            stage_operands(N - 2, sv, rf);
            mf_d4 cpre = stage_start(N - 2, sv, rf);
            if (N >= 3) stage_operands(N - 3, sv, rf);
            lds_f *tw = cm + (N - 2) * PLEN + U0;                                    // where t of the stage's knot goes
            for (int kn = N - 2; kn >= 0; --kn) {
                mf_d4 c = mf_mma(cf[S::F_MB0], p[0], cpre);                          // + [AmBKt; B^T] p
                if constexpr (XS == 2) c = mf_mma(cf[S::F_MB1], p[1], c);
                if (kn >= 1) {
                    cpre = stage_start(kn - 1, sv, rf);
                    if (kn >= 2) stage_operands(kn - 2, sv, rf);
                }
                p[0] = c[0], p[1] = c[1];
                if (ok2 && !conv) *tw = (float)c[2];                                 // (a converged instance keeps its feed-forward term)
                tw -= PLEN;
            }
        }
        __syncthreads();
        if (last) break;
    }

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
                double acc = gc64[T::O_F + r];
#pragma unroll
                for (int c = 0; c < NX; ++c) acc = fma(gc64[T::O_A + r * NX + c], s_plant[j * NX + c], acc);
#pragma unroll
                for (int a = 0; a < NU; ++a) acc = fma(gc64[T::O_B + r * NU + a], (double)vbu[0][a], acc);
                xn[r] = acc;
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                s_plant[j * NX + r] = xn[r];
                if (active) Pe->mpc_x[so * NX + r] = (float)xn[r];
            }
            if (active) {
#pragma unroll
                for (int a = 0; a < NU; ++a) Pe->mpc_u[so * NU + a] = vbu[0][a];
                Pe->mpc_iter[so] = conv ? it : -it;
            }
        }
        __syncthreads();
        x0r[0] = s_plant[j * NX + g];
        x0r[1] = ok1 ? s_plant[j * NX + row1] : 0.0;
        if (step + 1 < n_steps) continue;
        if (active) {
            Pe->x0_out[b * NX + g] = (float)x0r[0];
            if (ok1) Pe->x0_out[b * NX + row1] = (float)x0r[1];
        }
    }
    }   // closed-loop steps

    // the projected slack of the last executed iteration is the solution (admm.cpp:187-188,204-205)
    kparam_ptr Pe = kparams();
    mf_for<0, NG>([&](auto mt) {
        constexpr int m = decltype(mt)::value;
        const int kk = 4 * m + g;
        lds_f *c = cq + m * 4 * PLEN;
        if (active && kk < N) {
            float *po = Pe->xout + ox + m * 4 * NX;
#pragma unroll
            for (int r = 0; r < NX; ++r) po[r] = vbx[m][r];
            if (keep) {
                float *pg = Pe->sg + ox + m * 4 * NX;
#pragma unroll
                for (int r = 0; r < NX; ++r) pg[r] = a1x[m][r];
                if constexpr (CXQ > 0) {
                    float *pgc = Pe->sgc + ox + m * 4 * NX;
#pragma unroll
                    for (int c2 = 0; c2 < CXQ; ++c2) pgc[CXA + c2] = a2x[m][c2];
                }
                if (!conv) {                                   // (a converged instance parked the slack of the iteration before)
                    float *pv = Pe->sv + ox + m * 4 * NX;
#pragma unroll
                    for (int r = 0; r < NX; ++r) pv[r] = vbx[m][r];
                    if constexpr (CXQ > 0) {
                        float *pvc = Pe->svc + ox + m * 4 * NX;
#pragma unroll
                        for (int r = 0; r < NX; ++r) pvc[r] = vcx[m][r];
                    }
                }
            }
        }
        if (active && kk < N - 1) {
            float *po = Pe->uout + ou + m * 4 * NU;
#pragma unroll
            for (int a = 0; a < NU; ++a) po[a] = vbu[m][a];
            if (keep) {
                float *py = Pe->sy + ou + m * 4 * NU;
#pragma unroll
                for (int a = 0; a < NU; ++a) py[a] = a1u[m][a];
                if constexpr (CUQ > 0) {
                    float *pyc = Pe->syc + ou + m * 4 * NU;
#pragma unroll
                    for (int c2 = 0; c2 < CUQ; ++c2) pyc[CUA + c2] = a2u[m][c2];
                }
                if (!conv) {
                    float *pz = Pe->sz + ou + m * 4 * NU, *pd = Pe->sd + ou + m * 4 * NU;
#pragma unroll
                    for (int a = 0; a < NU; ++a) pz[a] = vbu[m][a];
                    if constexpr (CUQ > 0) {
                        float *pzc = Pe->szc + ou + m * 4 * NU;
#pragma unroll
                        for (int a = 0; a < NU; ++a) pzc[a] = vcu[m][a];
                    }
                    double tv[NU];                             // d = Quu_inv t of the last backward sweep (admm.cpp:17)
#pragma unroll
                    for (int a = 0; a < NU; ++a) tv[a] = (double)c[U0 + a * 16];
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        double acc = 0.0;
#pragma unroll
                        for (int a2 = 0; a2 < NU; ++a2) acc = fma(gc64[T::O_QUI + a * NU + a2], tv[a2], acc);
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

}  // namespace tmpc
