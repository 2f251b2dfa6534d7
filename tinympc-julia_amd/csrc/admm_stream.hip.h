// Fused ADMM kernel with a run-time horizon: "stream" layout.
//
// Same two fused sweeps as the quad kernel (admm_quad.hip.h; reference admm.cpp:13-207) but with N a
// run-time value, so ONE instantiation per (nx, nu) serves every horizon — including horizons whose
// trajectories cannot stay on chip (rocket N = 50 with cone slack pairs is 15 KB per instance).
//   * one lane per instance; the knot loops are rolled;
//   * the per-instance trajectories (g, v, vnew, y, z, znew, d [+ cone pairs]) stream through an HBM
//     scratch block laid out [element][batch] — every access is a fully coalesced 256-B line per
//     wavefront, and at the benchmark sizes the block (<= a few hundred MB) sits in the 256 MB
//     Infinity Cache / L2 rather than in HBM proper; the next knot's operands are loaded while the
//     current knot computes (explicit software pipelining: a lone wavefront has nothing else to hide
//     the load latency behind);
//   * coefficient rows are wave-uniform: staged once per workgroup in LDS and read as broadcasts;
//   * recurrences in RT (double by default), stored state and elementwise steps fp32 — same rule as
//     the quad kernel;
//   * carries the UNPINNED extensions (affine term, second-order cones) so BASELINE config 4 has a
//     fused path.
#pragma once
#include <hip/hip_runtime.h>

#include "admm_params.h"

namespace tmpc {

// coefficient pack (elements of RT, row-major rows): A[nx][nx] K[nu][nx] B[nx][nu] | AT[nx][nx]
// BT[nu][nx] KT[nx][nu] QI[nu][nu] | PT[nx][nx] | f[nx] APf[nx] BPf[nu]
template <int NX, int NU>
struct StreamPack {
    static constexpr int O_A = 0, O_K = O_A + NX * NX, O_B = O_K + NU * NX, O_AT = O_B + NX * NU,
                         O_BT = O_AT + NX * NX, O_KT = O_BT + NU * NX, O_QI = O_KT + NX * NU,
                         O_PT = O_QI + NU * NU, O_F = O_PT + NX * NX, O_APF = O_F + NX, O_BPF = O_APF + NX,
                         LEN = O_BPF + NU;
};
// bounds pack (fp32): [N][xmin[NX] xmax[NX] umin[NU] umax[NU]] then Qd[NX] Rd[NU]
// scratch arrays, each [knot][row][batch]:  g w v (x side), y zw z d (u side), [gc wc vc | yc zwc zc]

__device__ __forceinline__ float sfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double sfma(double a, double b, double c) { return fma(a, b, c); }

// Coefficient source: one family for the whole batch (LDS broadcast), or one family PER INSTANCE
// (SURVEY.md §8f-3): pack laid out [element][batch] in HBM, each lane reading its own column — the
// per-instance cache then streams from L2 / HBM (4(2nx^2 + 2nx nu + nu^2) B per knot and side).
template <class RT, bool HET>
struct CoefSrc {
    const RT *base;
    long stride;
    __device__ __forceinline__ RT operator[](int i) const { return HET ? base[(long)i * stride] : base[i]; }
    __device__ __forceinline__ CoefSrc operator+(int off) const { return CoefSrc{base + (HET ? (long)off * stride : off), stride}; }
};

template <int Q_MAX>
__device__ __forceinline__ void project_soc_regs(float (&blk)[Q_MAX], int first, int q, float mu) {
    // block rows [first, first+q), last one the axis: ||head|| <= mu * axis  (same map as the oracle)
    float a2 = 0.f;
#pragma unroll
    for (int j = 0; j < Q_MAX; ++j)
        if (j >= first && j < first + q - 1) a2 = fmaf(blk[j], blk[j], a2);
    float axis = 0.f;
#pragma unroll
    for (int j = 0; j < Q_MAX; ++j)
        if (j == first + q - 1) axis = blk[j];
    const float an = sqrtf(a2), u0 = axis * mu;
    const bool zero = an <= -u0, keep = !zero && an <= u0;
    const float sc = zero ? 0.f : (keep ? 1.f : 0.5f * (1.f + u0 / an));
    const float ax_new = zero ? 0.f : (keep ? axis : sc * (an / mu));
#pragma unroll
    for (int j = 0; j < Q_MAX; ++j) {
        if (j >= first && j < first + q - 1) blk[j] *= sc;
        if (j == first + q - 1) blk[j] = ax_new;
    }
}

template <int NX, int NU, class RT, bool EXT, bool HET>
__global__ __launch_bounds__(256) void admm_stream_kernel(const AdmmParams P) {
    using PK = StreamPack<NX, NU>;
    constexpr int T = 256;
    extern __shared__ __align__(16) unsigned char s_raw[];
    RT *s_coef = reinterpret_cast<RT *>(s_raw);
    float *s_bnd = reinterpret_cast<float *>(s_raw + sizeof(RT) * ((PK::LEN + 1) & ~1));

    const int N = P.N;
    const int tid = threadIdx.x;
    const RT *gcoef = reinterpret_cast<const RT *>(P.coef);
    if constexpr (!HET)
        for (int i = tid; i < PK::LEN; i += T) s_coef[i] = gcoef[i];
    const int bnd_len = N * (2 * NX + 2 * NU) + NX + NU;
    for (int i = tid; i < bnd_len; i += T) s_bnd[i] = P.bounds[i];
    __syncthreads();

    const long B = P.batch;
    const long b = (long)blockIdx.x * T + tid;
    const bool active = b < B;
    const long EX = (long)NX * N, EU = (long)NU * (N - 1);
    // per-family scalars: shared (bounds pack tail, kernel argument) or per instance (het_aux: [Qd|Rd|rho][batch])
    float cQD[NX], cRD[NU];
    float rho = P.rho;
    if constexpr (HET) {
        const long bb = active ? b : 0;
#pragma unroll
        for (int m = 0; m < NX; ++m) cQD[m] = P.het_aux[(long)m * B + bb];
#pragma unroll
        for (int a = 0; a < NU; ++a) cRD[a] = P.het_aux[(long)(NX + a) * B + bb];
        rho = P.het_aux[(long)(NX + NU) * B + bb];
    } else {
#pragma unroll
        for (int m = 0; m < NX; ++m) cQD[m] = s_bnd[N * (2 * NX + 2 * NU) + m];
#pragma unroll
        for (int a = 0; a < NU; ++a) cRD[a] = s_bnd[N * (2 * NX + 2 * NU) + NX + a];
    }
    const CoefSrc<RT, HET> cbase{HET ? gcoef + (active ? b : 0) : s_coef, HET ? B : 1};
    const auto cA = cbase + PK::O_A, cK = cbase + PK::O_K, cB = cbase + PK::O_B, cAT = cbase + PK::O_AT,
               cBT = cbase + PK::O_BT, cKT = cbase + PK::O_KT, cQI = cbase + PK::O_QI, cPT = cbase + PK::O_PT,
               cF = cbase + PK::O_F, cAPF = cbase + PK::O_APF, cBPF = cbase + PK::O_BPF;
    const bool soc_x = EXT && P.ncx > 0, soc_u = EXT && P.ncu > 0;

    // scratch columns of this instance
    float *Sg = P.scratch + (active ? b : 0), *Sw = Sg + EX * B, *Sv = Sw + EX * B;
    float *Sy = Sv + EX * B, *Szw = Sy + EU * B, *Sz = Szw + EU * B, *Sd = Sz + EU * B;
    float *Sgc = Sd + EU * B, *Swc = Sgc + EX * B, *Svc = Swc + EX * B;
    float *Syc = Svc + EX * B, *Szwc = Syc + EU * B, *Szc = Szwc + EU * B;
#define SX(arr, k, m) arr[((long)(k)*NX + (m)) * B]
#define SU(arr, k, m) arr[((long)(k)*NU + (m)) * B]

    RT x0[NX];
#pragma unroll
    for (int m = 0; m < NX; ++m) x0[m] = active ? (RT)P.x0[b * NX + m] : (RT)0;
    const bool warm = active && !P.cold_start;
    if (active) {
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int m = 0; m < NX; ++m) {
                SX(Sg, k, m) = warm ? P.sg[b * EX + k * NX + m] : 0.f;
                SX(Sv, k, m) = warm ? P.sv[b * EX + k * NX + m] : 0.f;
                SX(Sw, k, m) = 0.f;
                if (soc_x) {
                    SX(Sgc, k, m) = warm ? P.sgc[b * EX + k * NX + m] : 0.f;
                    SX(Svc, k, m) = warm ? P.svc[b * EX + k * NX + m] : 0.f;
                    SX(Swc, k, m) = 0.f;
                }
            }
        for (int k = 0; k < N - 1; ++k)
#pragma unroll
            for (int m = 0; m < NU; ++m) {
                SU(Sy, k, m) = warm ? P.sy[b * EU + k * NU + m] : 0.f;
                SU(Sz, k, m) = warm ? P.sz[b * EU + k * NU + m] : 0.f;
                SU(Sd, k, m) = warm ? P.sd[b * EU + k * NU + m] : 0.f;
                SU(Szw, k, m) = 0.f;
                if (soc_u) {
                    SU(Syc, k, m) = warm ? P.syc[b * EU + k * NU + m] : 0.f;
                    SU(Szc, k, m) = warm ? P.szc[b * EU + k * NU + m] : 0.f;
                    SU(Szwc, k, m) = 0.f;
                }
            }
    }
    auto ref_x = [&](int k, int m) -> float {
        if (P.ref_mode == REF_SHARED) return P.xref[k * NX + m];
        if (P.ref_mode == REF_PER_INSTANCE) return P.xref[b * EX + k * NX + m];
        return 0.f;
    };
    auto ref_u = [&](int k, int m) -> float {
        if (P.ref_mode == REF_SHARED) return P.uref[k * NU + m];
        if (P.ref_mode == REF_PER_INSTANCE) return P.uref[b * EU + k * NU + m];
        return 0.f;
    };

    int it = 0, conv = 0;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    if (warm) {
        res0 = P.res[b * 4 + 0];
        res1 = P.res[b * 4 + 1];
        res2 = P.res[b * 4 + 2];
        res3 = P.res[b * 4 + 3];
    }
    const int ct = P.check_termination;
    int ct_count = ct;
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;

    for (int i = 0; i < P.max_iter; ++i) {
        if (active && !conv) {
            bool check = false;
            if (ct > 0 && --ct_count == 0) {
                check = true;
                ct_count = ct;
            }
            const bool need_res = check && (can_converge || it + 1 == last_check_it);
            // ================= fused forward sweep (admm.cpp:25-69 + :93-96) =================
            RT x[NX];
#pragma unroll
            for (int m = 0; m < NX; ++m) x[m] = x0[m];
            float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
            // software pipeline: operands of knot k+1 are in flight while knot k computes
            float g_n[NX], v_n[NX], gc_n[NX], vc_n[NX], d_n[NU], y_n[NU], z_n[NU], yc_n[NU], zc_n[NU];
            auto fetch = [&](int k) {
#pragma unroll
                for (int m = 0; m < NX; ++m) {
                    g_n[m] = SX(Sg, k, m);
                    v_n[m] = need_res ? SX(Sv, k, m) : 0.f;
                    gc_n[m] = soc_x ? SX(Sgc, k, m) : 0.f;
                    vc_n[m] = (soc_x && need_res) ? SX(Svc, k, m) : 0.f;
                }
                if (k < N - 1) {
#pragma unroll
                    for (int m = 0; m < NU; ++m) {
                        d_n[m] = SU(Sd, k, m);
                        y_n[m] = SU(Sy, k, m);
                        z_n[m] = need_res ? SU(Sz, k, m) : 0.f;
                        yc_n[m] = soc_u ? SU(Syc, k, m) : 0.f;
                        zc_n[m] = (soc_u && need_res) ? SU(Szc, k, m) : 0.f;
                    }
                }
            };
            fetch(0);
            for (int k = 0; k < N; ++k) {
                float g_c[NX], v_c[NX], gc_c[NX], vc_c[NX], d_c[NU], y_c[NU], z_c[NU], yc_c[NU], zc_c[NU];
#pragma unroll
                for (int m = 0; m < NX; ++m) {
                    g_c[m] = g_n[m];
                    v_c[m] = v_n[m];
                    gc_c[m] = gc_n[m];
                    vc_c[m] = vc_n[m];
                }
#pragma unroll
                for (int m = 0; m < NU; ++m) {
                    d_c[m] = d_n[m];
                    y_c[m] = y_n[m];
                    z_c[m] = z_n[m];
                    yc_c[m] = yc_n[m];
                    zc_c[m] = zc_n[m];
                }
                if (k + 1 < N) fetch(k + 1);
                const float *bk = s_bnd + k * (2 * NX + 2 * NU);
                float xf[NX];
#pragma unroll
                for (int m = 0; m < NX; ++m) {
                    xf[m] = (float)x[m];
                    float vn = xf[m] + g_c[m];
                    vn = fminf(bk[NX + m], fmaxf(bk[m], vn));
                    SX(Sg, k, m) = (g_c[m] + xf[m]) - vn;
                    pri_x = fmaxf(pri_x, fabsf(xf[m] - vn));
                    dua_x = fmaxf(dua_x, fabsf(v_c[m] - vn));
                    SX(Sw, k, m) = vn;
                }
                if constexpr (EXT) {
                    if (soc_x) {
                        float wc[NX];
#pragma unroll
                        for (int m = 0; m < NX; ++m) wc[m] = xf[m] + gc_c[m];
                        for (int c = 0; c < P.ncx; ++c) project_soc_regs<NX>(wc, P.Acx[c], P.qcx[c], P.cx[c]);
#pragma unroll
                        for (int m = 0; m < NX; ++m) {
                            SX(Sgc, k, m) = (gc_c[m] + xf[m]) - wc[m];
                            pri_x = fmaxf(pri_x, fabsf(xf[m] - wc[m]));
                            dua_x = fmaxf(dua_x, fabsf(vc_c[m] - wc[m]));
                            SX(Swc, k, m) = wc[m];
                        }
                    }
                }
                if (k < N - 1) {
                    RT u[NU], xn[NX];
                    float uf[NU];
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        RT acc = 0;
#pragma unroll
                        for (int j = 0; j < NX; ++j) acc = sfma(cK[a * NX + j], x[j], acc);
                        u[a] = -acc - (RT)d_c[a];
                        uf[a] = (float)u[a];
                        float zn = uf[a] + y_c[a];
                        zn = fminf(bk[2 * NX + NU + a], fmaxf(bk[2 * NX + a], zn));
                        SU(Sy, k, a) = (y_c[a] + uf[a]) - zn;
                        pri_u = fmaxf(pri_u, fabsf(uf[a] - zn));
                        dua_u = fmaxf(dua_u, fabsf(z_c[a] - zn));
                        SU(Szw, k, a) = zn;
                    }
                    if constexpr (EXT) {
                        if (soc_u) {
                            float zc2[NU];
#pragma unroll
                            for (int a = 0; a < NU; ++a) zc2[a] = uf[a] + yc_c[a];
                            for (int c = 0; c < P.ncu; ++c) project_soc_regs<NU>(zc2, P.Acu[c], P.qcu[c], P.cu[c]);
#pragma unroll
                            for (int a = 0; a < NU; ++a) {
                                SU(Syc, k, a) = (yc_c[a] + uf[a]) - zc2[a];
                                pri_u = fmaxf(pri_u, fabsf(uf[a] - zc2[a]));
                                dua_u = fmaxf(dua_u, fabsf(zc_c[a] - zc2[a]));
                                SU(Szwc, k, a) = zc2[a];
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        RT acc = EXT ? cF[r] : (RT)0;
#pragma unroll
                        for (int j = 0; j < NX; ++j) acc = sfma(cA[r * NX + j], x[j], acc);
#pragma unroll
                        for (int a = 0; a < NU; ++a) acc = sfma(cB[r * NU + a], u[a], acc);
                        xn[r] = acc;
                    }
#pragma unroll
                    for (int r = 0; r < NX; ++r) x[r] = xn[r];
                }
            }
            it += 1;
            if (need_res) {
                res0 = pri_x;
                res1 = dua_x * rho;
                res2 = pri_u;
                res3 = dua_u * rho;
                if (res0 < P.abs_pri_tol && res2 < P.abs_pri_tol && res1 < P.abs_dua_tol && res3 < P.abs_dua_tol)
                    conv = 1;
            }
            if (!conv) {
                // ================= fused backward sweep (admm.cpp:75-83, :196-197, :13-20) =================
                RT p[NX];
                {
                    float wN[NX], gN[NX];
#pragma unroll
                    for (int m = 0; m < NX; ++m) {
                        wN[m] = SX(Sw, N - 1, m);
                        gN[m] = SX(Sg, N - 1, m);
                    }
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        RT acc = 0;
                        if (P.ref_mode != REF_ZERO) {
#pragma unroll
                            for (int j = 0; j < NX; ++j) acc = sfma(cPT[r * NX + j], (RT)ref_x(N - 1, j), acc);
                        }
                        float tail = rho * (wN[r] - gN[r]);
                        SX(Sv, N - 1, r) = wN[r];
                        if constexpr (EXT) {
                            if (soc_x) {
                                const float wc = SX(Swc, N - 1, r);
                                tail += rho * (wc - SX(Sgc, N - 1, r));
                                SX(Svc, N - 1, r) = wc;
                            }
                        }
                        p[r] = -acc - (RT)tail;
                    }
                }
                float w_n[NX], g_n2[NX], wc_n[NX], gc_n2[NX], zw_n[NU], y_n2[NU], zwc_n[NU], yc_n2[NU];
                auto fetchb = [&](int k) {
#pragma unroll
                    for (int m = 0; m < NX; ++m) {
                        w_n[m] = SX(Sw, k, m);
                        g_n2[m] = SX(Sg, k, m);
                        wc_n[m] = soc_x ? SX(Swc, k, m) : 0.f;
                        gc_n2[m] = soc_x ? SX(Sgc, k, m) : 0.f;
                    }
#pragma unroll
                    for (int m = 0; m < NU; ++m) {
                        zw_n[m] = SU(Szw, k, m);
                        y_n2[m] = SU(Sy, k, m);
                        zwc_n[m] = soc_u ? SU(Szwc, k, m) : 0.f;
                        yc_n2[m] = soc_u ? SU(Syc, k, m) : 0.f;
                    }
                };
                if (N >= 2) fetchb(N - 2);
                for (int k = N - 2; k >= 0; --k) {
                    RT r[NU], qk[NX];
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        float rr = P.ref_mode != REF_ZERO ? -(ref_u(k, a) * cRD[a]) : 0.f;
                        rr -= rho * (zw_n[a] - y_n2[a]);
                        SU(Sz, k, a) = zw_n[a];
                        if (soc_u) {
                            rr -= rho * (zwc_n[a] - yc_n2[a]);
                            SU(Szc, k, a) = zwc_n[a];
                        }
                        r[a] = (RT)rr;
                    }
#pragma unroll
                    for (int m = 0; m < NX; ++m) {
                        float qq = P.ref_mode != REF_ZERO ? -(ref_x(k, m) * cQD[m]) : 0.f;
                        qq -= rho * (w_n[m] - g_n2[m]);
                        SX(Sv, k, m) = w_n[m];
                        if (soc_x) {
                            qq -= rho * (wc_n[m] - gc_n2[m]);
                            SX(Svc, k, m) = wc_n[m];
                        }
                        qk[m] = (RT)qq;
                    }
                    if (k > 0) fetchb(k - 1);
                    RT t[NU];
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        RT acc = r[a] + (EXT ? cBPF[a] : (RT)0);
#pragma unroll
                        for (int j = 0; j < NX; ++j) acc = sfma(cBT[a * NX + j], p[j], acc);
                        t[a] = acc;
                    }
#pragma unroll
                    for (int a = 0; a < NU; ++a) {
                        RT acc = 0;
#pragma unroll
                        for (int c = 0; c < NU; ++c) acc = sfma(cQI[a * NU + c], t[c], acc);
                        SU(Sd, k, a) = (float)acc;
                    }
                    RT pn[NX];
#pragma unroll
                    for (int m = 0; m < NX; ++m) {
                        RT ap = qk[m] + (EXT ? cAPF[m] : (RT)0), kr = 0;
#pragma unroll
                        for (int j = 0; j < NX; ++j) ap = sfma(cAT[m * NX + j], p[j], ap);
#pragma unroll
                        for (int a = 0; a < NU; ++a) kr = sfma(cKT[m * NU + a], r[a], kr);
                        pn[m] = ap - kr;
                    }
#pragma unroll
                    for (int m = 0; m < NX; ++m) p[m] = pn[m];
                }
            }
        }
        if (!__builtin_amdgcn_ballot_w64(active && !conv)) break;
    }

    if (active) {
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int m = 0; m < NX; ++m) P.xout[b * EX + k * NX + m] = SX(Sw, k, m);
        for (int k = 0; k < N - 1; ++k)
#pragma unroll
            for (int m = 0; m < NU; ++m) P.uout[b * EU + k * NU + m] = SU(Szw, k, m);
        P.iter[b] = it;
        P.solved[b] = conv;
        P.res[b * 4 + 0] = res0;
        P.res[b * 4 + 1] = res1;
        P.res[b * 4 + 2] = res2;
        P.res[b * 4 + 3] = res3;
        if (P.save_state) {
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int m = 0; m < NX; ++m) {
                    P.sg[b * EX + k * NX + m] = SX(Sg, k, m);
                    P.sv[b * EX + k * NX + m] = SX(Sv, k, m);
                    if (soc_x) {
                        P.sgc[b * EX + k * NX + m] = SX(Sgc, k, m);
                        P.svc[b * EX + k * NX + m] = SX(Svc, k, m);
                    }
                }
            for (int k = 0; k < N - 1; ++k)
#pragma unroll
                for (int m = 0; m < NU; ++m) {
                    P.sy[b * EU + k * NU + m] = SU(Sy, k, m);
                    P.sz[b * EU + k * NU + m] = SU(Sz, k, m);
                    P.sd[b * EU + k * NU + m] = SU(Sd, k, m);
                    if (soc_u) {
                        P.syc[b * EU + k * NU + m] = SU(Syc, k, m);
                        P.szc[b * EU + k * NU + m] = SU(Szc, k, m);
                    }
                }
        }
    }
#undef SX
#undef SU
    {
        float m0 = active ? res0 : 0.f, m1 = active ? res1 : 0.f, m2 = active ? res2 : 0.f, m3 = active ? res3 : 0.f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
            m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
        }
        const unsigned long long unsolved = __builtin_amdgcn_ballot_w64(active && !conv);
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
