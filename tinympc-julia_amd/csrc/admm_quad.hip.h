// Fused TinyMPC ADMM kernel for gfx950 — "quad" layout.
//
// What it computes: the whole of the reference's solve() loop
//   forward_pass -> update_slack -> update_dual -> update_linear_cost ->
//   termination_condition -> (v,z = vnew,znew) -> backward_pass_grad
//   (reference: src/codegen_src/tinympc/admm.cpp:109-207, phases :13-107)
// for a batch of independent problem instances that share one problem family
// (A, B, Q, R, rho, bounds), with per-instance x0 / references / warm-start state.
//
// Mapping (MI355X, wave64):
//   * 4 lanes — one DPP quad — per problem instance, 16 instances per wavefront,
//     64 per 256-thread workgroup.  Lane q of the quad owns state rows
//     [q*RX, (q+1)*RX) and input rows [q*RU, (q+1)*RU), RX = ceil(nx/4), RU = ceil(nu/4).
//     (One whole wavefront per instance, as first sketched in the north star, leaves
//     >= 52 of 64 lanes idle in the serial Riccati/rollout recurrences where only nx
//     rows of work exist per step; quads keep every lane busy for nx in {4, 12} and
//     75 % for nx = 6.  See DESIGN.md.)
//   * the small mat-vecs (nx x nx, nu x nx, ...) are row-per-lane FMAs whose vector
//     operand is fetched from the owning lane of the quad with a DPP quad_perm
//     broadcast — no LDS, no ds_bpermute, no MFMA (4x4 .. 12x12 is not a contraction
//     worth a matrix core).
//   * every trajectory the ADMM iteration carries (g, v, vnew, y, z, znew, d) lives in
//     VGPRs for the whole solve; the horizon loops are fully unrolled so the arrays are
//     statically indexed.  x, u, q, r, p of the reference are never materialised: they
//     are produced and consumed knot by knot inside the two fused sweeps.
//   * family constants: coefficient rows are loaded once per lane; per-knot bounds (and
//     shared references) are staged once per workgroup in LDS.
//   * HBM traffic is the compulsory I/O only: x0 (+refs, +warm state) in, x/u/status
//     (+warm state) out.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "admm_params.h"

namespace tmpc {

template <int NX_, int NU_, int N_>
struct QuadShape {
    static constexpr int NX = NX_, NU = NU_, N = N_;
    static constexpr int RX = (NX + 3) / 4, RU = (NU + 3) / 4;
    static constexpr int NXP = 4 * RX, NUP = 4 * RU;
    static constexpr int NXL = (NX + RX - 1) / RX;  // lanes of a quad owning real x rows
    static constexpr int NUL = (NU + RU - 1) / RU;  // lanes of a quad owning real u rows
    // Coefficient pack per lane role q (floats); rows beyond nx/nu and columns beyond
    // nx/nu are zero.  Filled by host (solver.cpp: build_quad_coef).
    static constexpr int O_A = 0;                 // A      rows [RX][NXP]
    static constexpr int O_AT = O_A + RX * NXP;   // AmBKt  rows [RX][NXP]
    static constexpr int O_K = O_AT + RX * NXP;   // Kinf   rows [RU][NXP]
    static constexpr int O_B = O_K + RU * NXP;    // B      rows [RX][NUP]
    static constexpr int O_BT = O_B + RX * NUP;   // B^T    rows [RU][NXP]
    static constexpr int O_KT = O_BT + RU * NXP;  // Kinf^T rows [RX][NUP]
    static constexpr int O_QI = O_KT + RX * NUP;  // Quu_inv rows [RU][NUP]
    static constexpr int O_PT = O_QI + RU * NUP;  // Pinf^T rows [RX][NXP]
    static constexpr int O_QD = O_PT + RX * NXP;  // diag(Q)+rho [RX]
    static constexpr int O_RD = O_QD + RX;        // diag(R)+rho [RU]
    static constexpr int CP = O_RD + RU;
    // Bounds pack: [N][4 roles][xmin[RX] xmax[RX] umin[RU] umax[RU]]
    static constexpr int BW = 2 * RX + 2 * RU;
    static constexpr int BOUNDS_LEN = N * 4 * BW;
    // Shared-reference pack: [N][4 roles][xref[RX] uref[RU]]
    static constexpr int RW = RX + RU;
    static constexpr int REFS_LEN = N * 4 * RW;
    static constexpr int INST_PER_BLOCK = 64;
    static constexpr int THREADS = 256;
};

// ---- compile-time loop, so DPP controls are integer constant expressions ----
template <int I, int E, class F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, E>(f);
    }
}

// DPP quad_perm: lane l reads lane (l & ~3) | perm[l & 3] of its own quad.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int S>
__device__ __forceinline__ float qbcast(float v) {
    return dpp_quad<S * 0x55>(v);  // quad_perm:[S,S,S,S]
}
__device__ __forceinline__ float quad_max(float m) {
    m = fmaxf(m, dpp_quad<0xB1>(m));  // quad_perm:[1,0,3,2]
    m = fmaxf(m, dpp_quad<0x4E>(m));  // quad_perm:[2,3,0,1]
    return m;
}

// acc[m] += sum_{s < SL} sum_{k < SK} C[m*CW + s*SK + k] * (src[k] of quad lane s)
template <int ROWS, int SL, int SK, int CW>
__device__ __forceinline__ void quad_matvec(float (&acc)[ROWS], const float *C,
                                            const float (&src)[SK]) {
    sfor<0, SL>([&](auto s) {
        constexpr int S = decltype(s)::value;
#pragma unroll
        for (int k = 0; k < SK; ++k) {
            const float xv = qbcast<S>(src[k]);
#pragma unroll
            for (int m = 0; m < ROWS; ++m) acc[m] = fmaf(C[m * CW + S * SK + k], xv, acc[m]);
        }
    });
}

template <class S, int REFS>
__global__ __launch_bounds__(256) void admm_quad_kernel(const AdmmParams P) {
    constexpr int NX = S::NX, NU = S::NU, N = S::N;
    constexpr int RX = S::RX, RU = S::RU, NXP = S::NXP, NUP = S::NUP;
    constexpr int NXL = S::NXL, NUL = S::NUL;
    constexpr int EX = NX * N, EU = NU * (N - 1);

    __shared__ float s_bnd[S::BOUNDS_LEN];
    __shared__ float s_ref[REFS == REF_SHARED ? S::REFS_LEN : 1];

    const int tid = threadIdx.x;
    for (int i = tid; i < S::BOUNDS_LEN; i += S::THREADS) s_bnd[i] = P.bounds[i];
    if constexpr (REFS == REF_SHARED) {
        // pack [N][4][xref[RX] uref[RU]] from row-major-per-knot xref [N][nx], uref [N-1][nu]
        for (int i = tid; i < S::REFS_LEN; i += S::THREADS) {
            const int k = i / (4 * S::RW), rem = i % (4 * S::RW);
            const int qq = rem / S::RW, j = rem % S::RW;
            float val = 0.f;
            if (j < RX) {
                const int row = qq * RX + j;
                if (row < NX) val = P.xref[k * NX + row];
            } else {
                const int row = qq * RU + (j - RX);
                if (row < NU && k < N - 1) val = P.uref[k * NU + row];
            }
            s_ref[i] = val;
        }
    }
    __syncthreads();

    const int q = tid & 3;
    const long b = (long)blockIdx.x * S::INST_PER_BLOCK + (tid >> 2);
    const bool active = b < P.batch;
    const float *lb = s_bnd + q * S::BW;
    const float *lr = s_ref + q * S::RW;

    // ---- per-lane coefficient rows ----
    const float *cp = P.coef + q * S::CP;
    float cA[RX * NXP], cAT[RX * NXP], cK[RU * NXP], cB[RX * NUP], cBT[RU * NXP], cKT[RX * NUP],
        cQI[RU * NUP], cQD[RX], cRD[RU];
#pragma unroll
    for (int i = 0; i < RX * NXP; ++i) cA[i] = cp[S::O_A + i];
#pragma unroll
    for (int i = 0; i < RX * NXP; ++i) cAT[i] = cp[S::O_AT + i];
#pragma unroll
    for (int i = 0; i < RU * NXP; ++i) cK[i] = cp[S::O_K + i];
#pragma unroll
    for (int i = 0; i < RX * NUP; ++i) cB[i] = cp[S::O_B + i];
#pragma unroll
    for (int i = 0; i < RU * NXP; ++i) cBT[i] = cp[S::O_BT + i];
#pragma unroll
    for (int i = 0; i < RX * NUP; ++i) cKT[i] = cp[S::O_KT + i];
#pragma unroll
    for (int i = 0; i < RU * NUP; ++i) cQI[i] = cp[S::O_QI + i];
#pragma unroll
    for (int i = 0; i < RX; ++i) cQD[i] = cp[S::O_QD + i];
#pragma unroll
    for (int i = 0; i < RU; ++i) cRD[i] = cp[S::O_RD + i];
    const float rho = P.rho;

    // ---- per-instance state, all in registers ----
    float g[N][RX], v[N][RX], w[N][RX];
    float y[N - 1][RU], z[N - 1][RU], zw[N - 1][RU], d[N - 1][RU];
    float x0[RX];
    float xr[REFS == REF_PER_INSTANCE ? N : 1][RX];
    float ur[REFS == REF_PER_INSTANCE ? N - 1 : 1][RU];

#pragma unroll
    for (int m = 0; m < RX; ++m) {
        const int row = q * RX + m;
        x0[m] = (active && row < NX) ? P.x0[b * NX + row] : 0.f;
    }
    const bool warm = active && !P.cold_start;
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int m = 0; m < RX; ++m) {
            const int row = q * RX + m;
            const bool ld = warm && row < NX;
            g[k][m] = ld ? P.sg[b * EX + k * NX + row] : 0.f;
            v[k][m] = ld ? P.sv[b * EX + k * NX + row] : 0.f;
            w[k][m] = 0.f;
            if constexpr (REFS == REF_PER_INSTANCE)
                xr[k][m] = (active && row < NX) ? P.xref[b * EX + k * NX + row] : 0.f;
        }
#pragma unroll
    for (int k = 0; k < N - 1; ++k)
#pragma unroll
        for (int m = 0; m < RU; ++m) {
            const int row = q * RU + m;
            const bool ld = warm && row < NU;
            y[k][m] = ld ? P.sy[b * EU + k * NU + row] : 0.f;
            z[k][m] = ld ? P.sz[b * EU + k * NU + row] : 0.f;
            d[k][m] = ld ? P.sd[b * EU + k * NU + row] : 0.f;
            zw[k][m] = 0.f;
            if constexpr (REFS == REF_PER_INSTANCE)
                ur[k][m] = (active && row < NU) ? P.uref[b * EU + k * NU + row] : 0.f;
        }

    auto ref_x = [&](auto kk, int m) -> float {
        constexpr int K = decltype(kk)::value;
        if constexpr (REFS == REF_SHARED) return lr[K * 4 * S::RW + m];
        else if constexpr (REFS == REF_PER_INSTANCE) return xr[K][m];
        else return 0.f;
    };
    auto ref_u = [&](auto kk, int m) -> float {
        constexpr int K = decltype(kk)::value;
        if constexpr (REFS == REF_SHARED) return lr[K * 4 * S::RW + RX + m];
        else if constexpr (REFS == REF_PER_INSTANCE) return ur[K][m];
        else return 0.f;
    };

    // admm.cpp:112-115 — only counters are reset at entry; the workspace persists.
    int it = 0;
    int conv = 0;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    if (warm) {
        // residual fields persist in the reference's workspace too (types.hpp:128-131)
        res0 = P.res[b * 4 + 0];
        res1 = P.res[b * 4 + 1];
        res2 = P.res[b * 4 + 2];
        res3 = P.res[b * 4 + 3];
    }
    const int ct = P.check_termination;
    int ct_count = ct;

    for (int i = 0; i < P.max_iter; ++i) {
        if (active && !conv) {
            // ================= fused forward sweep =================
            // forward_pass (admm.cpp:25-35) + update_slack (:43-59) + update_dual (:65-69)
            // + the residual maxima of termination_condition (:93-96), knot by knot.
            float x[RX];
#pragma unroll
            for (int m = 0; m < RX; ++m) x[m] = x0[m];
            float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
#pragma unroll
            for (int k = 0; k < N; ++k) {
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    float vn = x[m] + g[k][m];                                  // vnew = x + g
                    vn = fminf(lb[k * 4 * S::BW + RX + m],                      // x_max.cwiseMin(
                               fmaxf(lb[k * 4 * S::BW + m], vn));               //   x_min.cwiseMax(vnew))
                    g[k][m] = (g[k][m] + x[m]) - vn;                            // g = g + x - vnew
                    pri_x = fmaxf(pri_x, fabsf(x[m] - vn));
                    dua_x = fmaxf(dua_x, fabsf(v[k][m] - vn));
                    w[k][m] = vn;
                }
                if (k < N - 1) {
                    float u[RU];
#pragma unroll
                    for (int m = 0; m < RU; ++m) u[m] = 0.f;
                    quad_matvec<RU, NXL, RX, NXP>(u, cK, x);                    // Kinf x
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        u[m] = -u[m] - d[k][m];                                 // u = -Kinf x - d
                        float zn = u[m] + y[k][m];                              // znew = u + y
                        zn = fminf(lb[k * 4 * S::BW + 2 * RX + RU + m],
                                   fmaxf(lb[k * 4 * S::BW + 2 * RX + m], zn));
                        y[k][m] = (y[k][m] + u[m]) - zn;                        // y = y + u - znew
                        pri_u = fmaxf(pri_u, fabsf(u[m] - zn));
                        dua_u = fmaxf(dua_u, fabsf(z[k][m] - zn));
                        zw[k][m] = zn;
                    }
                    float xn[RX], bu[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) xn[m] = bu[m] = 0.f;
                    quad_matvec<RX, NXL, RX, NXP>(xn, cA, x);                   // A x
                    quad_matvec<RX, NUL, RU, NUP>(bu, cB, u);                   // B u
#pragma unroll
                    for (int m = 0; m < RX; ++m) x[m] = xn[m] + bu[m];
                }
            }
            it += 1;  // admm.cpp:143

            // ================= termination_condition (admm.cpp:89-107) =================
            bool check = false;
            if (ct > 0) {
                if (--ct_count == 0) {
                    check = true;
                    ct_count = ct;
                }
            }
            if (check) {
                res0 = quad_max(pri_x);
                res1 = quad_max(dua_x) * rho;
                res2 = quad_max(pri_u);
                res3 = quad_max(dua_u) * rho;
                if (res0 < P.abs_pri_tol && res2 < P.abs_pri_tol && res1 < P.abs_dua_tol &&
                    res3 < P.abs_dua_tol)
                    conv = 1;
            }
            if (!conv) {
                // ================= fused backward sweep =================
                // v = vnew, z = znew (admm.cpp:196-197); update_linear_cost (:75-83) and
                // backward_pass_grad (:13-20) knot by knot, q/r/p never stored.
                float p[RX];
                {
                    float acc[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) acc[m] = 0.f;
                    if constexpr (REFS != REF_ZERO) {
                        float xrl[RX];
#pragma unroll
                        for (int m = 0; m < RX; ++m)
                            xrl[m] = ref_x(std::integral_constant<int, N - 1>{}, m);
                        // -(Xref_{N-1}^T Pinf)^T  (admm.cpp:81)
                        float cPT[RX * NXP];
#pragma unroll
                        for (int i2 = 0; i2 < RX * NXP; ++i2) cPT[i2] = cp[S::O_PT + i2];
                        quad_matvec<RX, NXL, RX, NXP>(acc, cPT, xrl);
                    }
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        p[m] = -acc[m] - rho * (w[N - 1][m] - g[N - 1][m]);    // admm.cpp:81-82
                        v[N - 1][m] = w[N - 1][m];
                    }
                }
                sfor<0, N - 1>([&](auto kk) {
                    constexpr int k = N - 2 - decltype(kk)::value;
                    constexpr std::integral_constant<int, k> kc{};
                    float r[RU], qk[RX];
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        float rr = 0.f;
                        if constexpr (REFS != REF_ZERO) {
                            rr = -(ref_u(kc, m) * cRD[m]);                               // -(Uref .* R)
                        }
                        r[m] = rr - rho * (zw[k][m] - y[k][m]);                 // admm.cpp:77-78
                        z[k][m] = zw[k][m];
                    }
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        float qq = 0.f;
                        if constexpr (REFS != REF_ZERO) {
                            qq = -(ref_x(kc, m) * cQD[m]);                               // -(Xref .* Q)
                        }
                        qk[m] = qq - rho * (w[k][m] - g[k][m]);                 // admm.cpp:79-80
                        v[k][m] = w[k][m];
                    }
                    float t[RU];
#pragma unroll
                    for (int m = 0; m < RU; ++m) t[m] = 0.f;
                    quad_matvec<RU, NXL, RX, NXP>(t, cBT, p);                   // B^T p_{k+1}
#pragma unroll
                    for (int m = 0; m < RU; ++m) t[m] += r[m];                  //   + r_k
                    float dn[RU];
#pragma unroll
                    for (int m = 0; m < RU; ++m) dn[m] = 0.f;
                    quad_matvec<RU, NUL, RU, NUP>(dn, cQI, t);                  // d_k = Quu_inv (...)
#pragma unroll
                    for (int m = 0; m < RU; ++m) d[k][m] = dn[m];
                    float ap[RX], kr[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) ap[m] = kr[m] = 0.f;
                    quad_matvec<RX, NXL, RX, NXP>(ap, cAT, p);                  // AmBKt p_{k+1}
                    quad_matvec<RX, NUL, RU, NUP>(kr, cKT, r);                  // Kinf^T r_k
#pragma unroll
                    for (int m = 0; m < RX; ++m) p[m] = (qk[m] + ap[m]) - kr[m];  // admm.cpp:18
                });
            }
        }
        // every instance of this wavefront finished?  (wave-uniform exit)
        if (!__builtin_amdgcn_ballot_w64(active && !conv)) break;
    }

    // ================= epilogue: solution, status, warm-start state =================
    if (active) {
        // solution = projected slack of the last executed iteration (admm.cpp:187-188,204-205)
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                if (row < NX) P.xout[b * EX + k * NX + row] = w[k][m];
            }
#pragma unroll
        for (int k = 0; k < N - 1; ++k)
#pragma unroll
            for (int m = 0; m < RU; ++m) {
                const int row = q * RU + m;
                if (row < NU) P.uout[b * EU + k * NU + row] = zw[k][m];
            }
        if (q == 0) {
            P.iter[b] = it;
            P.solved[b] = conv;
            P.res[b * 4 + 0] = res0;
            P.res[b * 4 + 1] = res1;
            P.res[b * 4 + 2] = res2;
            P.res[b * 4 + 3] = res3;
        }
        if (P.save_state) {
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    const int row = q * RX + m;
                    if (row < NX) {
                        P.sg[b * EX + k * NX + row] = g[k][m];
                        P.sv[b * EX + k * NX + row] = v[k][m];
                    }
                }
#pragma unroll
            for (int k = 0; k < N - 1; ++k)
#pragma unroll
                for (int m = 0; m < RU; ++m) {
                    const int row = q * RU + m;
                    if (row < NU) {
                        P.sy[b * EU + k * NU + row] = y[k][m];
                        P.sz[b * EU + k * NU + row] = z[k][m];
                        P.sd[b * EU + k * NU + row] = d[k][m];
                    }
                }
        }
    }
    // global status block: wavefront max of the residuals, count of unsolved instances
    {
        float m0 = active ? res0 : 0.f, m1 = active ? res1 : 0.f, m2 = active ? res2 : 0.f,
              m3 = active ? res3 : 0.f;
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) {
            m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
            m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
        }
        const unsigned long long unsolved = __builtin_amdgcn_ballot_w64(active && !conv && q == 0);
        if ((tid & 63) == 0) {
            atomicMax(&P.gstat[0], __float_as_uint(m0));
            atomicMax(&P.gstat[1], __float_as_uint(m1));
            atomicMax(&P.gstat[2], __float_as_uint(m2));
            atomicMax(&P.gstat[3], __float_as_uint(m3));
            const int n = __popcll(unsolved);
            if (n) atomicAdd(&P.gstat[4], (uint32_t)n);
        }
    }
}

}  // namespace tmpc
