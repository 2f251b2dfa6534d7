// Fused ADMM kernel with the recurrences on the fp64 matrix cores, the per-instance state in LDS, rolled knot loops:
// "mfmac<nx,nu>" — run-time horizon, box bounds + affine dynamics term + second-order cones (BASELINE config 4).
//
// Why: the run-time-horizon stream kernel (admm_streamg.hip.h) keeps every trajectory in HBM and moves ~12 KB per
// instance and ADMM iteration through it (40 GB per launch for config 4: 340 x the algorithmic bytes).  Only the
// instances that are being iterated need their state on chip, and a one-shot solve needs little of it:
//   per knot and row   one dual per constraint set (g | y, gc | yc),
//                      ONE fused array handed from the forward to the backward sweep: sum over sets of (slack - dual),
//                      which for input rows shares its slot with the feed-forward d (d_k is written by the backward
//                      sweep exactly where su_k was read, and read by the forward sweep before su_k is written)
// = 24 floats per knot for the rocket (6 state rows, 3 of them in a cone, 3 input rows in a cone): 4.7 KB per instance
// at N = 50, so 16 instances — one wavefront — take 75 KB of LDS and a CU holds two wavefronts' worth.  Nothing of it
// ever goes to HBM: traffic is x0 in, the solution out, plus (see "residuals") the previous slack around a check.
//
// Mapping: one wavefront = 16 instances (the N side of a 16x16x4 fp64 MFMA tile), lane l = 16 g + j works on instance
// j.  The stacked vector [x; u] occupies tile rows / K indices  x_c -> c (c < 8),  u_a -> 8 + a, i.e. with the
// instruction's layout (A: lane l holds A[l % 16][l / 16]; B: B[l / 16][l % 16]; D: register v of lane l holds
// D[4 v + l / 16][l % 16], experiments/mfma_probe.hip) lane group g carries in its three "slots"
//     slot 0: x_g      slot 1: x_{4+g}      slot 2: u_g
// and a product's result registers are the next product's B operands as they stand: no cross-lane move on the chain.
//     forward : c = {f, nd};  c += M_u nd;  c += M_x0 x[0];  c += M_x1 x[1]      -> c[0..1] = x+,  c[2] = u = -Kinf x - d
//               (M = [A - B Kinf, B; -Kinf, 0], nd = -d; the product with nd does not wait for x)
//     backward: c = {q + APf, r + BPf};  c += N_u r;  c += N_x0 p[0];  c += N_x1 p[1]   -> c[0..1] = p-,  c[2] = B'p + r
//               (N = [AmBKt, -Kinf'; B', 0]);  d = Quu_inv c[2] is one more product, off the chain
// The x slots of position k hold knot k + 1 (the rollout produces x_{k+1} together with u_k), so both sweeps index the
// state with ONE wave-uniform position; the backward sweep reads position i for q_{i+1} (used at once) and r_i (used one
// stage later).  Knot 0 of the state side (x0 is given) is handled once per iteration outside the loop.
//
// Cones may span lane groups (their rows are consecutive components): squared head norms and the axis value are
// summed over the four lanes of an instance (two xor-shuffles each), off the recurrence chain.
//
// Residuals (termination_condition, admm.cpp:89-107) need the PREVIOUS iteration's slack of every set, which this
// layout does not keep.  On an iteration that precedes a check the forward sweep therefore also writes the box slack to
// the solution buffers (xout / uout — where the slack of the last executed iteration has to end up anyway) and the cone
// slack to an HBM scratch, and the checking iteration reads them back: once per solve for the fixed-iteration benchmark
// configs, every check_termination-th iteration otherwise.
//
// Scope: one-shot solves (cold start, workspace not kept), shared or zero references, at most 8 state / 4 input rows,
// cones and box bounds as the C-ABI takes them; no linear-inequality rows, no per-instance families, no adaptive rho
// (those stay on the stream / generic kernels).  Precision as everywhere: fp64 recurrences, fp32 state.
#pragma once
#include <hip/hip_runtime.h>

#include "admm_params.h"
#include "admm_mfma.hip.h"

namespace tmpc {

template <int NX, int NU>
struct ConeShape {
    static_assert(NX >= 1 && NX <= 8 && NU >= 1 && NU <= 4, "mfmac kernel: nx <= 8, nu <= 4");
    static constexpr int XS = NX > 4 ? 2 : 1;  // state slots in use
    // fp64 operand fields, [field][64 lanes]
    enum { F_MF0 = 0, F_MF1, F_MF2, F_MB0, F_MB1, F_MB2, F_MQ, F_FD0, F_FD1, F_APF0, F_APF1, F_BPF, NF };
    static constexpr int NROW = NX + NU;
    // fp32 pack: `nk` knots of [lo(NROW) hi(NROW)] (state rows of knot k, input rows of knot k; nk = 1 when the bounds do
    // not depend on the knot, AdmmParams::bounds_stride = 0), then Qd[NX] Rd[NU], then -inf, +inf pads
    static constexpr int bounds_len(int nk) { return 2 * NROW * nk + NROW + 2; }
    // LDS state, per position (a knot's worth of the 16 instances): [A1: NROW rows][A3: NROW rows][A2: rows that lie in
    // a cone] x 16 instances, + one zero pad cell per lane for the slots a lane does not own
    static constexpr int pos_len(int cone_rows) { return 16 * (2 * NROW + cone_rows); }
    static constexpr size_t lds_bytes(int N, int nk, int cone_rows) {
        return sizeof(float) * ((size_t)pos_len(cone_rows) * (N - 1) + 64 + bounds_len(nk) + (((size_t)NROW * N + 1) & ~(size_t)1)) +
               sizeof(double) * 8;
    }
    // HBM scratch per wavefront (floats): cone slack of the iteration before a check, [pos 0..N-1][slot][lane]
    static constexpr size_t scratch_floats(int N) { return (size_t)N * 3 * 64; }
};

// sum over the four lanes (16 apart) of an instance
__device__ __forceinline__ float mfc_inst_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int NX, int NU, int REFS, bool EXT>
__global__ __launch_bounds__(64) void admm_mfmac_kernel(const AdmmParams P) {
    using S = ConeShape<NX, NU>;
    constexpr int XS = S::XS, NROW = S::NROW;
    extern __shared__ __align__(16) unsigned char s_raw_c[];
    const int N = P.N;
    const int ncx = EXT ? P.ncx : 0, ncu = EXT ? P.ncu : 0;
    // rows that lie in some cone get a cone dual of their own (A2); every other row's stays identically zero
    auto in_cone = [&](int rho) -> bool {
        if (rho < NX) {
            for (int c = 0; c < ncx; ++c)
                if (rho >= P.Acx[c] && rho < P.Acx[c] + P.qcx[c]) return true;
        } else {
            for (int c = 0; c < ncu; ++c)
                if (rho - NX >= P.Acu[c] && rho - NX < P.Acu[c] + P.qcu[c]) return true;
        }
        return false;
    };
    int cone_rows = 0;
    for (int rho = 0; rho < NROW; ++rho) cone_rows += in_cone(rho) ? 1 : 0;
    const int PLEN = S::pos_len(cone_rows);
    const int nk = P.bounds_stride ? N : 1;
    float *s_state = reinterpret_cast<float *>(s_raw_c);
    float *s_pad = s_state + (size_t)PLEN * (N - 1);
    float *s_bnd = s_pad + 64;
    float *s_ref = s_bnd + S::bounds_len(nk);
    double *s_pterm = reinterpret_cast<double *>(s_ref + (((size_t)NROW * N + 1) & ~(size_t)1));
    __shared__ uint4 s_cmask[8 * 4];  // [cone][lane group]: x head bits, x axis bits, u head bit, u axis bit (bit = slot)

    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const long slot_id = (long)blockIdx.x * 16 + j;
    const bool active = slot_id < P.batch;
    const long b = active ? slot_id : 0;
    const long EX = (long)NX * N, EU = (long)NU * (N - 1);
    // rows of this lane: slot 0 -> x_g, slot 1 -> x_{4+g}, slot 2 -> u_g
    const int row0 = g, row1 = 4 + g, row2 = g;
    const bool ok0 = row0 < NX, ok1 = row1 < NX, ok2 = row2 < NU;

    // ---- stage constants ----
    for (int i = l; i < S::bounds_len(nk); i += 64) s_bnd[i] = P.bounds[i];
    const float *Qd = s_bnd + 2 * NROW * nk, *Rd = Qd + NX;
    if constexpr (REFS == REF_SHARED) {
        // -(Xref .* Q~), -(Uref .* R~) as update_linear_cost forms them (admm.cpp:77-80), per knot
        for (int i = l; i < NROW * N; i += 64) {
            const int k = i / NROW, r = i % NROW;
            float v = 0.f;
            if (r < NX) v = -(P.xref[k * NX + r] * P.bounds[2 * NROW * nk + r]);
            else if (k < N - 1) v = -(P.uref[k * NU + (r - NX)] * P.bounds[2 * NROW * nk + r]);
            s_ref[i] = v;
        }
    }
    const double *gc64 = reinterpret_cast<const double *>(P.coef);
    double cf[S::NF];
#pragma unroll
    for (int f = 0; f < S::NF; ++f) cf[f] = gc64[f * 64 + l];
    if constexpr (REFS == REF_SHARED) {
        // terminal cost: -(Xref_{N-1}' Pinf)' (admm.cpp:81-82); Pinf follows the lane fields, row-major [NX][NX]
        const double *Pinf = gc64 + S::NF * 64;
        if (l < NX) {
            double acc = 0.0;
            for (int c = 0; c < NX; ++c) acc = fma(Pinf[c * NX + l], (double)P.xref[(N - 1) * NX + c], acc);  // (Pinf^T xref)[l]
            s_pterm[l] = -acc;
        }
    }
    if constexpr (EXT) {
        if (l < 32) {
            const int c = l >> 2, gg = l & 3;
            unsigned hx = 0u, ax = 0u, hu = 0u, au = 0u;
            if (c < ncx)
                for (int sl = 0; sl < 2; ++sl) {
                    const int row = 4 * sl + gg;
                    if (row < NX && row >= P.Acx[c] && row < P.Acx[c] + P.qcx[c] - 1) hx |= 1u << sl;
                    if (row < NX && row == P.Acx[c] + P.qcx[c] - 1) ax |= 1u << sl;
                }
            if (c < ncu) {
                if (gg < NU && gg >= P.Acu[c] && gg < P.Acu[c] + P.qcu[c] - 1) hu = 1u;
                if (gg < NU && gg == P.Acu[c] + P.qcu[c] - 1) au = 1u;
            }
            s_cmask[l] = make_uint4(hx, ax, hu, au);
        }
    }
    // cold start = the zero workspace tiny_setup leaves (tiny_api.cpp:73-88)
    for (int i = l; i < PLEN * (N - 1) + 64; i += 64) s_state[i] = 0.f;
    __syncthreads();

    const bool soc_x = ncx > 0, soc_u = ncu > 0;
    const float rho = P.rho;
    const float qd0 = ok0 ? Qd[row0] : 0.f, qd1 = ok1 ? Qd[row1] : 0.f, rd2 = ok2 ? Rd[row2] : 0.f;
    (void)qd0, (void)qd1, (void)rd2;
    // bounds of this lane's rows at knot k: s_bnd[k * 2 NROW + row] (lo), [+ NROW] (hi); rows that do not exist -> pads
    const int PAD_LO = 2 * NROW * nk + NROW, PAD_HI = PAD_LO + 1, BST = P.bounds_stride ? 2 * NROW : 0;
    if (l == 0) {
        s_bnd[PAD_LO] = -__builtin_inff();
        s_bnd[PAD_HI] = __builtin_inff();
    }
    __syncthreads();
    const int bx0 = ok0 ? row0 : -1, bx1 = ok1 ? row1 : -1, bu2 = ok2 ? NX + row2 : -1;
    auto lo_of = [&](int k, int br) -> float { return br < 0 ? s_bnd[PAD_LO] : s_bnd[k * BST + br]; };
    auto hi_of = [&](int k, int br) -> float { return br < 0 ? s_bnd[PAD_HI] : s_bnd[k * BST + NROW + br]; };
    // state accessors: element (array, slot) of position `pos` for this lane; slots the lane does not own (and cone duals
    // of rows outside every cone) go to the lane's pad cell, which holds an exact zero for the whole solve
    int s_off[3][3], s_str[3][3];
    {
        const int rhox[3] = {row0, row1, NX + row2};      // row index in the stacked [x; u] numbering
        const bool okr[3] = {ok0, ok1, ok2};
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            int crow = 0;
            for (int r2 = 0; r2 < rhox[sl] && r2 < NROW; ++r2) crow += in_cone(r2) ? 1 : 0;
            const bool cone = okr[sl] && in_cone(rhox[sl]);
            s_off[0][sl] = okr[sl] ? rhox[sl] * 16 + j : PLEN * (N - 1) + l;
            s_off[2][sl] = okr[sl] ? (NROW + rhox[sl]) * 16 + j : PLEN * (N - 1) + l;
            s_off[1][sl] = cone ? (2 * NROW + crow) * 16 + j : PLEN * (N - 1) + l;
            s_str[0][sl] = s_str[2][sl] = okr[sl] ? PLEN : 0;
            s_str[1][sl] = cone ? PLEN : 0;
        }
    }
    auto S_ = [&](int pos, int arr, int sl) -> float & { return s_state[pos * s_str[arr][sl] + s_off[arr][sl]]; };
    // HBM scratch of this wavefront: cone slack kept around a check
    float *const scr = P.scratch + (size_t)blockIdx.x * S::scratch_floats(N) + l;
    auto SCR = [&](int pos, int sl) -> float & { return scr[((size_t)pos * 3 + sl) * 64]; };

    double x0r[2];
    x0r[0] = (active && ok0) ? (double)P.x0[b * NX + row0] : 0.0;
    x0r[1] = (active && ok1) ? (double)P.x0[b * NX + row1] : 0.0;
    float g0[2] = {0.f, 0.f}, gc0[2] = {0.f, 0.f};  // duals of knot 0 (state side)

    int it = 0, conv = 0;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    const int ct = P.check_termination;
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;

    // One projection pass over the cone set of a side.  v[]: this lane's slots of the side (x: 2, u: 1), in place.
    auto project_x = [&](float (&v)[2]) {
        for (int c = 0; c < ncx; ++c) {
            const uint4 mk = s_cmask[c * 4 + g];
            float a2 = 0.f, axv = 0.f;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if ((mk.x >> sl) & 1u) a2 = fmaf(v[sl], v[sl], a2);
                if ((mk.y >> sl) & 1u) axv = v[sl];
            }
            a2 = mfc_inst_sum(a2);
            axv = mfc_inst_sum(axv);
            const float mu = P.cx[c], an = sqrtf(a2), u0 = axv * mu;
            const bool zero = an <= -u0, keep = !zero && an <= u0;
            const float sc = zero ? 0.f : (keep ? 1.f : 0.5f * (1.f + u0 / an));
            const float ax_new = zero ? 0.f : (keep ? axv : sc * (an / mu));
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if ((mk.x >> sl) & 1u) v[sl] *= sc;
                if ((mk.y >> sl) & 1u) v[sl] = ax_new;
            }
        }
    };
    auto project_u = [&](float &v) {
        for (int c = 0; c < ncu; ++c) {
            const uint4 mk = s_cmask[c * 4 + g];
            float a2 = (mk.z & 1u) ? v * v : 0.f, axv = (mk.w & 1u) ? v : 0.f;
            a2 = mfc_inst_sum(a2);
            axv = mfc_inst_sum(axv);
            const float mu = P.cu[c], an = sqrtf(a2), u0 = axv * mu;
            const bool zero = an <= -u0, keep = !zero && an <= u0;
            const float sc = zero ? 0.f : (keep ? 1.f : 0.5f * (1.f + u0 / an));
            const float ax_new = zero ? 0.f : (keep ? axv : sc * (an / mu));
            if (mk.z & 1u) v *= sc;
            if (mk.w & 1u) v = ax_new;
        }
    };

    for (int i = 0; i < P.max_iter; ++i) {
        const int itn = i + 1;
        const bool check = ct > 0 && itn % ct == 0;
        const bool need_res = check && (can_converge || itn == last_check_it);
        const bool check_next = ct > 0 && (itn + 1) % ct == 0 && itn < P.max_iter;
        const bool write_old = check_next && (can_converge || itn + 1 == last_check_it);   // the next iteration reads this one's slack
        const bool last = itn == P.max_iter;
        const bool write_sol = write_old || need_res || last;   // box slack -> xout / uout
        const bool read_old = need_res && itn > 1;               // the zero workspace before the first iteration
        const bool wr = active && !conv;                         // a converged instance's outputs are frozen
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;

        // state side of one knot: slack / dual of both sets for this lane's two state slots.  xv: the knot's state (fp64),
        // kn: knot index, A1 / A2: duals in, out through the references; returns the fused (slack - dual) sum per slot.
        auto state_knot = [&](const double (&xv)[2], int kn, int spos, float (&a1)[2], float (&a2)[2], float (&sx)[2]) {
            float xf[2], vn[2], vc[2];
            const int brow[2] = {bx0, bx1};
            const bool okr[2] = {ok0, ok1};
#pragma unroll
            for (int sl = 0; sl < XS; ++sl) {
                xf[sl] = (float)xv[sl];
                vn[sl] = fminf(hi_of(kn, brow[sl]), fmaxf(lo_of(kn, brow[sl]), xf[sl] + a1[sl]));   // admm.cpp:52-56
                a1[sl] = (a1[sl] + xf[sl]) - vn[sl];                                                // admm.cpp:68
                sx[sl] = vn[sl] - a1[sl];
            }
            if constexpr (XS == 1) xf[1] = vn[1] = vc[1] = sx[1] = 0.f;
            if (need_res) {
#pragma unroll
                for (int sl = 0; sl < XS; ++sl) {
                    float old = 0.f;
                    if (read_old && active && okr[sl]) old = P.xout[b * EX + (long)kn * NX + (sl ? row1 : row0)];
                    pri_x = fmaxf(pri_x, fabsf(xf[sl] - vn[sl]));
                    dua_x = fmaxf(dua_x, fabsf(old - vn[sl]));
                }
            }
            if (write_sol && wr) {
#pragma unroll
                for (int sl = 0; sl < XS; ++sl)
                    if (okr[sl]) P.xout[b * EX + (long)kn * NX + (sl ? row1 : row0)] = vn[sl];
            }
            if constexpr (EXT) {
                if (soc_x) {
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) vc[sl] = sl < XS ? xf[sl] + a2[sl] : 0.f;
                    project_x(vc);
#pragma unroll
                    for (int sl = 0; sl < XS; ++sl) {
                        a2[sl] = (a2[sl] + xf[sl]) - vc[sl];
                        sx[sl] += vc[sl] - a2[sl];
                    }
                    if (need_res) {
#pragma unroll
                        for (int sl = 0; sl < XS; ++sl) {
                            const float old = read_old ? SCR(spos, sl) : 0.f;
                            pri_x = fmaxf(pri_x, fabsf(xf[sl] - vc[sl]));
                            dua_x = fmaxf(dua_x, fabsf(old - vc[sl]));
                        }
                    }
                    if (write_old) {
#pragma unroll
                        for (int sl = 0; sl < XS; ++sl) SCR(spos, sl) = vc[sl];
                    }
                }
            }
        };

        // ================= fused forward sweep (admm.cpp:25-69, :93-96) =================
        {
            float sx0[2];
            state_knot(x0r, 0, N - 1, g0, gc0, sx0);      // knot 0: its fused value feeds nothing (q_0 only enters p_0)
        }
        double x[2] = {x0r[0], x0r[1]};
        // operands of position 0
        float nA1x[2], nA2x[2], nA1u, nA2u, nd_f;
        nA1x[0] = S_(0, 0, 0), nA1x[1] = S_(0, 0, 1), nA1u = S_(0, 0, 2);
        nA2x[0] = S_(0, 1, 0), nA2x[1] = S_(0, 1, 1), nA2u = S_(0, 1, 2);
        nd_f = S_(0, 2, 2);
        for (int k = 0; k < N - 1; ++k) {
            float a1x[2] = {nA1x[0], nA1x[1]}, a2x[2] = {nA2x[0], nA2x[1]};
            float a1u = nA1u, a2u = nA2u;
            const double nd = -(double)nd_f;
            mf_d4 c = {cf[S::F_FD0], cf[S::F_FD1], nd, 0.0};
            c = mf_mma(cf[S::F_MF2], nd, c);                                     // [B; 0] (-d): does not wait for x
            c = mf_mma(cf[S::F_MF0], x[0], c);                                   // + [A - B Kinf; -Kinf] x
            if constexpr (XS == 2) c = mf_mma(cf[S::F_MF1], x[1], c);
            if (k + 1 < N - 1) {                                                 // next position's operands, while the products run
                nA1x[0] = S_(k + 1, 0, 0), nA1x[1] = S_(k + 1, 0, 1), nA1u = S_(k + 1, 0, 2);
                nA2x[0] = S_(k + 1, 1, 0), nA2x[1] = S_(k + 1, 1, 1), nA2u = S_(k + 1, 1, 2);
                nd_f = S_(k + 1, 2, 2);
            }
            x[0] = c[0], x[1] = c[1];
            // input row of knot k
            {
                const float uf = (float)c[2];
                float zn = fminf(hi_of(k, bu2), fmaxf(lo_of(k, bu2), uf + a1u));
                a1u = (a1u + uf) - zn;
                float su = zn - a1u;
                if (need_res) {
                    float old = 0.f;
                    if (read_old && active && ok2) old = P.uout[b * EU + (long)k * NU + row2];
                    pri_u = fmaxf(pri_u, fabsf(uf - zn));
                    dua_u = fmaxf(dua_u, fabsf(old - zn));
                }
                if (write_sol && wr && ok2) P.uout[b * EU + (long)k * NU + row2] = zn;
                if constexpr (EXT) {
                    if (soc_u) {
                        float zc = uf + a2u;
                        project_u(zc);
                        a2u = (a2u + uf) - zc;
                        su += zc - a2u;
                        if (need_res) {
                            const float old = read_old ? SCR(k, 2) : 0.f;
                            pri_u = fmaxf(pri_u, fabsf(uf - zc));
                            dua_u = fmaxf(dua_u, fabsf(old - zc));
                        }
                        if (write_old) SCR(k, 2) = zc;
                    }
                }
                S_(k, 0, 2) = a1u;
                S_(k, 1, 2) = a2u;
                S_(k, 2, 2) = su;
            }
            // state rows of knot k + 1
            {
                float sx[2];
                state_knot(x, k + 1, k, a1x, a2x, sx);
                S_(k, 0, 0) = a1x[0], S_(k, 0, 1) = a1x[1];
                S_(k, 1, 0) = a2x[0], S_(k, 1, 1) = a2x[1];
                S_(k, 2, 0) = sx[0], S_(k, 2, 1) = sx[1];
            }
        }
        it += 1;
        if (need_res) {
            const float r0 = mf_inst_max(pri_x), r1 = mf_inst_max(dua_x) * rho, r2 = mf_inst_max(pri_u),
                        r3 = mf_inst_max(dua_u) * rho;
            if (!conv) {
                res0 = r0, res1 = r1, res2 = r2, res3 = r3;
                if (res0 < P.abs_pri_tol && res2 < P.abs_pri_tol && res1 < P.abs_dua_tol && res3 < P.abs_dua_tol) {
                    conv = 1;
                    if (active && g == 0) {
                        P.iter[b] = P.iter_offset + it;
                        P.solved[b] = 1;
                    }
                }
            }
        }
        if (last || !__builtin_amdgcn_ballot_w64(active && !conv)) break;
        // ================= fused backward sweep (admm.cpp:75-83, :13-20) =================
        double p[2], r_held;
        {
            const int pos = N - 2;
            double pt0 = 0.0, pt1 = 0.0;
            if constexpr (REFS == REF_SHARED) {
                pt0 = ok0 ? s_pterm[row0] : 0.0;
                pt1 = ok1 ? s_pterm[row1] : 0.0;
            }
            p[0] = pt0 - (double)(rho * S_(pos, 2, 0));                          // admm.cpp:81-82
            p[1] = pt1 - (double)(rho * S_(pos, 2, 1));
            float rr = 0.f;
            if constexpr (REFS == REF_SHARED) rr = ok2 ? s_ref[pos * NROW + NX + row2] : 0.f;
            r_held = (double)(rr - rho * S_(pos, 2, 2));                         // admm.cpp:77-78
        }
        for (int i2 = N - 3; i2 >= -1; --i2) {
            const int kk = i2 + 1;                                               // the knot this stage produces p and d of
            float q0 = 0.f, q1 = 0.f, rn = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
            if (i2 >= 0) {
                s0 = S_(i2, 2, 0), s1 = S_(i2, 2, 1), s2 = S_(i2, 2, 2);
                if constexpr (REFS == REF_SHARED) {
                    q0 = ok0 ? s_ref[kk * NROW + row0] : 0.f;
                    q1 = ok1 ? s_ref[kk * NROW + row1] : 0.f;
                    rn = ok2 ? s_ref[i2 * NROW + NX + row2] : 0.f;
                }
            }
            mf_d4 c = {(double)(q0 - rho * s0) + cf[S::F_APF0], (double)(q1 - rho * s1) + cf[S::F_APF1],
                       r_held + cf[S::F_BPF], 0.0};
            c = mf_mma(cf[S::F_MB2], r_held, c);                                 // [-Kinf^T; 0] r: does not wait for p
            c = mf_mma(cf[S::F_MB0], p[0], c);                                   // + [AmBKt; B^T] p
            if constexpr (XS == 2) c = mf_mma(cf[S::F_MB1], p[1], c);
            p[0] = c[0], p[1] = c[1];
            mf_d4 dq = {0.0, 0.0, 0.0, 0.0};
            dq = mf_mma(cf[S::F_MQ], c[2], dq);                                  // d = Quu_inv (B^T p + r), off the chain
            S_(kk, 2, 2) = (float)dq[2];
            r_held = (double)(rn - rho * s2);
        }
    }

    if (active && !conv && g == 0) {
        P.iter[b] = P.iter_offset + it;
        P.solved[b] = 0;
    }
    if (active && g == 0) {
        P.res[b * 4 + 0] = res0;
        P.res[b * 4 + 1] = res1;
        P.res[b * 4 + 2] = res2;
        P.res[b * 4 + 3] = res3;
    }
    {
        float m0 = active ? res0 : 0.f, m1 = active ? res1 : 0.f, m2 = active ? res2 : 0.f, m3 = active ? res3 : 0.f;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
            m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
        }
        const unsigned long long unsolved = __builtin_amdgcn_ballot_w64(active && !conv && g == 0);
        fold_status(P, m0, m1, m2, m3, __popcll(unsolved), l);
    }
}

}  // namespace tmpc
