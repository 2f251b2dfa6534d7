// Fused ADMM kernel with a run-time horizon, 4 lanes (one DPP quad) per instance: "stream4".
//
// The stream kernel's idea (admm_stream.hip.h: rolled knot loops, trajectories streamed through an
// L2 / Infinity-Cache-resident scratch block, next-knot operands in flight while the current knot
// computes) on the quad kernel's lane mapping (admm_quad.hip.h: lane q of a quad owns state rows
// [q*RX,(q+1)*RX) and input rows [q*RU,(q+1)*RU), mat-vec operands fetched with DPP quad_perm
// broadcasts).  Against one lane per instance it puts 4x the wavefronts in flight — what a
// latency-bound streaming kernel needs at the benchmark batch sizes (32 768 rocket instances are only
// 512 single-lane wavefronts for 1 024 SIMDs) — and a third of the work per lane.
//   * scratch layout [array][knot][4*batch + lane][local rows]: a lane's rows are one access, a wavefront's
//     accesses one contiguous span;
//   * one family for the batch: coefficient rows per lane role in LDS (conflict-free image, as in the
//     quad kernel); one family per instance (HET): the same rows as per-lane columns in HBM;
//   * second-order cones may straddle lanes: squared head norms and the axis value are summed over
//     the quad with two DPP steps.
#pragma once
#include <hip/hip_runtime.h>

#include "admm_params.h"
#include "admm_quad.hip.h"
#include "admm_stream.hip.h"

namespace tmpc {

// Pack: the quad kernel's three blocks (QuadShape offsets; N plays no role in them) + the affine block
template <int NX, int NU>
struct Stream4Pack {
    using S = QuadShape<NX, NU, 2, 4>;
    static constexpr int RX = S::RX, RU = S::RU;
    static constexpr int pad4(int n) { return (n + 3) / 4 * 4; }  // CoefLds offsets are whole 16-byte chunks
    static constexpr int O_F = S::CP, O_APF = O_F + pad4(RX), O_BPF = O_APF + pad4(RX);
    static constexpr int CP = S::pad8(O_BPF + RU);
    static constexpr int BW = 2 * RX + 2 * RU;  // bounds per knot and role: xmin xmax umin umax
    static constexpr int DW = RX + RU;          // diag(Q)+rho, diag(R)+rho per role
};

// the (wave-uniform) knot index, hidden from loop strength reduction
__device__ __forceinline__ int knot_sgpr(int k) {
    asm volatile("" : "+s"(k));
    return k;
}

// a wave-uniform pointer pinned to SGPRs: accesses become (SGPR base) + (32-bit VGPR offset)
template <class T>
__device__ __forceinline__ T *sgpr_ptr(T *p) {
    asm("" : "+s"(p));
    return p;
}

// uniform base + 32-bit byte offset of the lane: the (SGPR pair) + (VGPR) addressing mode of global_load/store.
// The explicit global address space keeps these from degrading to flat accesses once the pointer has been
// through sgpr_ptr's asm.
template <class T>
__device__ __forceinline__ auto lane_elem(T *uniform_base, unsigned byte_off) {
    using GC = std::conditional_t<std::is_const<T>::value, const char, char> __attribute__((address_space(1)));
    using GT = T __attribute__((address_space(1)));
    asm("" : "+v"(byte_off));  // keeps the 32->64-bit extension next to the access, where the addressing mode can absorb it
    return (GT *)((GC *)uniform_base + byte_off);
}

// One family per instance: element i of this lane's pack is column `lane` of row i of a [CP][4*batch] matrix.
template <class RT>
struct CoefCol {
    const RT *base;  // wave-uniform
    long stride;
    unsigned lane;  // byte offset of this lane's column
    __device__ __forceinline__ RT operator[](int i) const { return *lane_elem(sgpr_ptr(base + i * stride), lane); }
    __device__ __forceinline__ CoefCol operator+(int off) const { return CoefCol{base + off * stride, stride, lane}; }
};

__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_quad<0xB1>(v);
    v += dpp_quad<0x4E>(v);
    return v;
}

// cone c restricted to this lane's R local rows: bit m of head / axis set when local row m belongs to it
template <int R>
__device__ __forceinline__ void project_soc_quad(float (&blk)[R], unsigned head, unsigned axis, float mu) {
    float a2 = 0.f, ax = 0.f;
#pragma unroll
    for (int m = 0; m < R; ++m) {
        if ((head >> m) & 1u) a2 = fmaf(blk[m], blk[m], a2);
        if ((axis >> m) & 1u) ax = blk[m];
    }
    a2 = quad_sum(a2);
    ax = quad_sum(ax);  // exactly one lane contributes
    const float an = sqrtf(a2), u0 = ax * mu;
    const bool zero = an <= -u0, keep = !zero && an <= u0;
    const float sc = zero ? 0.f : (keep ? 1.f : 0.5f * (1.f + u0 / an));
    const float ax_new = zero ? 0.f : (keep ? ax : sc * (an / mu));
#pragma unroll
    for (int m = 0; m < R; ++m) {
        if ((head >> m) & 1u) blk[m] *= sc;
        if ((axis >> m) & 1u) blk[m] = ax_new;
    }
}

#ifndef TMPC_STREAM4_WAVES
#define TMPC_STREAM4_WAVES 3  // wavefronts per SIMD the register allocation is held to (168 VGPRs)
#endif

template <int NX, int NU, class RT, bool EXT, bool HET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TMPC_STREAM4_WAVES))) void admm_stream4_kernel(const AdmmParams P) {
    using PK = Stream4Pack<NX, NU>;
    using S = typename PK::S;
    constexpr int G = 4, T = 256, RX = S::RX, RU = S::RU, NXP = S::NXP, NUP = S::NUP, NXL = S::NXL, NUL = S::NUL;
    extern __shared__ __align__(16) unsigned char s_raw4[];
    RT *s_coef = reinterpret_cast<RT *>(s_raw4);
    float *s_bnd = reinterpret_cast<float *>(s_raw4 + sizeof(RT) * G * PK::CP);
    __shared__ uint4 s_cmask[8 * G];

    const int N = P.N;
    const int tid = threadIdx.x;
    const RT *gcoef = reinterpret_cast<const RT *>(P.coef);
    if constexpr (!HET)  // role-major in HBM -> role-interleaved 16-byte chunks in LDS
        for (int i = tid; i < G * PK::CP; i += T) s_coef[CoefLds<RT, G>::slot(i % PK::CP, i / PK::CP)] = gcoef[i];
    const int bnd_len = N * G * PK::BW + G * PK::DW;
    for (int i = tid; i < bnd_len; i += T) s_bnd[i] = P.bounds[i];
    __syncthreads();

    const int q = tid & 3;
    const long B = P.batch, B4 = 4 * B;
    const long b = (long)blockIdx.x * (T / G) + tid / G;
    const bool active = b < B;
    const long L = active ? 4 * b + q : q;  // this lane's column in every [element][4*batch] array
    const long EX = (long)NX * N, EU = (long)NU * (N - 1);
    const float *lb = s_bnd + q * PK::BW;

    float cQD[RX], cRD[RU];
    float rho = P.rho;
    if constexpr (HET) {  // per instance: het_aux = [Qd (nx) | Rd (nu) | rho][batch]
        const long bb = active ? b : 0;
#pragma unroll
        for (int m = 0; m < RX; ++m) cQD[m] = (q * RX + m < NX) ? P.het_aux[(long)(q * RX + m) * B + bb] : 0.f;
#pragma unroll
        for (int m = 0; m < RU; ++m) cRD[m] = (q * RU + m < NU) ? P.het_aux[(long)(NX + q * RU + m) * B + bb] : 0.f;
        rho = P.het_aux[(long)(NX + NU) * B + bb];
    } else {
        const float *ld = s_bnd + N * G * PK::BW + q * PK::DW;
#pragma unroll
        for (int m = 0; m < RX; ++m) cQD[m] = ld[m];
#pragma unroll
        for (int m = 0; m < RU; ++m) cRD[m] = ld[RX + m];
    }
    using CPtr = std::conditional_t<HET, CoefCol<RT>, CoefLds<RT, G>>;
    CPtr cbase;
    if constexpr (HET)
        cbase = CoefCol<RT>{gcoef, B4, (unsigned)(L * sizeof(RT))};
    else
        cbase = CoefLds<RT, G>{s_coef + q * CoefLds<RT, G>::VEC};
    const CPtr cA = cbase + S::O_A, cK = cbase + S::O_K, cB = cbase + S::O_B, cAT = cbase + S::O_AT,
               cBT = cbase + S::O_BT, cKT = cbase + S::O_KT, cQI = cbase + S::O_QI, cPT = cbase + S::O_PT,
               cF = cbase + PK::O_F, cAPF = cbase + PK::O_APF, cBPF = cbase + PK::O_BPF;

    // cone membership of each role's local rows, as bit masks in LDS: [cone][role]{x heads, x axis, u heads, u axis}
    const int ncx = EXT ? P.ncx : 0, ncu = EXT ? P.ncu : 0;
    if constexpr (EXT) {
        if (tid < 8 * G) {
            const int c = tid / G, r = tid % G;
            unsigned hx = 0u, ax = 0u, hu = 0u, au = 0u;
            if (c < ncx)
                for (int m = 0; m < RX; ++m) {
                    const int row = r * RX + m;
                    if (row >= P.Acx[c] && row < P.Acx[c] + P.qcx[c] - 1) hx |= 1u << m;
                    if (row == P.Acx[c] + P.qcx[c] - 1) ax |= 1u << m;
                }
            if (c < ncu)
                for (int m = 0; m < RU; ++m) {
                    const int row = r * RU + m;
                    if (row >= P.Acu[c] && row < P.Acu[c] + P.qcu[c] - 1) hu |= 1u << m;
                    if (row == P.Acu[c] + P.qcu[c] - 1) au |= 1u << m;
                }
            s_cmask[tid] = make_uint4(hx, ax, hu, au);
        }
        __syncthreads();
    }
    const uint4 *cm = s_cmask + q;
    const bool soc_x = ncx > 0, soc_u = ncu > 0;

    // scratch: [array][knot][4*batch + lane][local rows] — a lane's rows of one knot are contiguous (one
    // 4..16-byte access per array and knot) and a wavefront covers one contiguous 64*R*4-byte span.
    // Addresses are formed as (uniform 64-bit base, SGPRs) + (32-bit lane offset, one VGPR for all arrays);
    // the knot index is made opaque per knot (knot_sgpr) so that no per-array 64-bit pointers are carried
    // through the sweeps in VGPRs.
    const long SXN = (long)RX * N * B4, SUN = (long)RU * (N - 1) * B4;
    const unsigned LX = (unsigned)(L * RX * 4), LU = (unsigned)(L * RU * 4);  // byte offsets of this lane
    float *const Sg = P.scratch, *const Sw = Sg + SXN, *const Sv = Sw + SXN;
    float *const Sy = Sv + SXN, *const Szw = Sy + SUN, *const Sz = Szw + SUN, *const Sd = Sz + SUN;
    float *const Sgc = Sd + SUN, *const Swc = Sgc + SXN, *const Svc = Swc + SXN;
    float *const Syc = Svc + SXN, *const Szwc = Syc + SUN, *const Szc = Szwc + SUN;
#define SX(arr, k, m) (*lane_elem(sgpr_ptr(arr + ((long)(k)*B4 * RX + (m))), LX))
#define SU(arr, k, m) (*lane_elem(sgpr_ptr(arr + ((long)(k)*B4 * RU + (m))), LU))

    RT x0[RX];
#pragma unroll
    for (int m = 0; m < RX; ++m) x0[m] = (active && q * RX + m < NX) ? (RT)P.x0[b * NX + q * RX + m] : (RT)0;
    const bool warm = active && !P.cold_start;
    if (active) {
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                const bool ld = warm && row < NX;
                SX(Sg, k, m) = ld ? P.sg[b * EX + k * NX + row] : 0.f;
                SX(Sv, k, m) = ld ? P.sv[b * EX + k * NX + row] : 0.f;
                SX(Sw, k, m) = 0.f;
                if (soc_x) {
                    SX(Sgc, k, m) = ld ? P.sgc[b * EX + k * NX + row] : 0.f;
                    SX(Svc, k, m) = ld ? P.svc[b * EX + k * NX + row] : 0.f;
                    SX(Swc, k, m) = 0.f;
                }
            }
        for (int k = 0; k < N - 1; ++k)
#pragma unroll
            for (int m = 0; m < RU; ++m) {
                const int row = q * RU + m;
                const bool ld = warm && row < NU;
                SU(Sy, k, m) = ld ? P.sy[b * EU + k * NU + row] : 0.f;
                SU(Sz, k, m) = ld ? P.sz[b * EU + k * NU + row] : 0.f;
                SU(Sd, k, m) = ld ? P.sd[b * EU + k * NU + row] : 0.f;
                SU(Szw, k, m) = 0.f;
                if (soc_u) {
                    SU(Syc, k, m) = ld ? P.syc[b * EU + k * NU + row] : 0.f;
                    SU(Szc, k, m) = ld ? P.szc[b * EU + k * NU + row] : 0.f;
                    SU(Szwc, k, m) = 0.f;
                }
            }
    }
    auto ref_x = [&](int k, int m) -> float {
        const int row = q * RX + m;
        if (row >= NX || P.ref_mode == REF_ZERO) return 0.f;
        return P.ref_mode == REF_SHARED ? P.xref[k * NX + row] : P.xref[b * EX + k * NX + row];
    };
    auto ref_u = [&](int k, int m) -> float {
        const int row = q * RU + m;
        if (row >= NU || P.ref_mode == REF_ZERO) return 0.f;
        return P.ref_mode == REF_SHARED ? P.uref[k * NU + row] : P.uref[b * EU + k * NU + row];
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
            RT x[RX];
#pragma unroll
            for (int m = 0; m < RX; ++m) x[m] = x0[m];
            float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
            float g_n[RX], v_n[RX], gc_n[RX], vc_n[RX], d_n[RU], y_n[RU], z_n[RU], yc_n[RU], zc_n[RU];
            auto fetch = [&](int k_) {
                const int k = knot_sgpr(k_);
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    g_n[m] = SX(Sg, k, m);
                    v_n[m] = need_res ? SX(Sv, k, m) : 0.f;
                    gc_n[m] = soc_x ? SX(Sgc, k, m) : 0.f;
                    vc_n[m] = (soc_x && need_res) ? SX(Svc, k, m) : 0.f;
                }
                if (k < N - 1) {
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        d_n[m] = SU(Sd, k, m);
                        y_n[m] = SU(Sy, k, m);
                        z_n[m] = need_res ? SU(Sz, k, m) : 0.f;
                        yc_n[m] = soc_u ? SU(Syc, k, m) : 0.f;
                        zc_n[m] = (soc_u && need_res) ? SU(Szc, k, m) : 0.f;
                    }
                }
            };
            fetch(0);
            for (int k_ = 0; k_ < N; ++k_) {
                asm volatile("" ::: "memory");  // keep coefficient / bound loads per knot (no hoisting into registers)
                const int k = knot_sgpr(k_);
                float g_c[RX], v_c[RX], gc_c[RX], vc_c[RX], d_c[RU], y_c[RU], z_c[RU], yc_c[RU], zc_c[RU];
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    g_c[m] = g_n[m];
                    v_c[m] = v_n[m];
                    gc_c[m] = gc_n[m];
                    vc_c[m] = vc_n[m];
                }
#pragma unroll
                for (int m = 0; m < RU; ++m) {
                    d_c[m] = d_n[m];
                    y_c[m] = y_n[m];
                    z_c[m] = z_n[m];
                    yc_c[m] = yc_n[m];
                    zc_c[m] = zc_n[m];
                }
                if (k + 1 < N) fetch(k + 1);
                const float *bk = lb + k * G * PK::BW;
                float xf[RX];
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    xf[m] = (float)x[m];
                    float vn = xf[m] + g_c[m];
                    vn = fminf(bk[RX + m], fmaxf(bk[m], vn));
                    SX(Sg, k, m) = (g_c[m] + xf[m]) - vn;
                    pri_x = fmaxf(pri_x, fabsf(xf[m] - vn));
                    dua_x = fmaxf(dua_x, fabsf(v_c[m] - vn));
                    SX(Sw, k, m) = vn;
                }
                if constexpr (EXT) {
                    if (soc_x) {
                        float wc[RX];
#pragma unroll
                        for (int m = 0; m < RX; ++m) wc[m] = xf[m] + gc_c[m];
                        for (int c = 0; c < ncx; ++c) {
                            const uint4 mk = cm[c * G];
                            project_soc_quad<RX>(wc, mk.x, mk.y, P.cx[c]);
                        }
#pragma unroll
                        for (int m = 0; m < RX; ++m) {
                            SX(Sgc, k, m) = (gc_c[m] + xf[m]) - wc[m];
                            pri_x = fmaxf(pri_x, fabsf(xf[m] - wc[m]));
                            dua_x = fmaxf(dua_x, fabsf(vc_c[m] - wc[m]));
                            SX(Swc, k, m) = wc[m];
                        }
                    }
                }
                if (k < N - 1) {
                    RT u[RU], xn[RX];
                    float uf[RU];
#pragma unroll
                    for (int m = 0; m < RU; ++m) u[m] = (RT)0;
#pragma unroll
                    for (int m = 0; m < RX; ++m) xn[m] = EXT ? (RT)cF[m] : (RT)0;
                    asm volatile("" ::: "memory");  // (and between products: each one's coefficient loads stay next to their use)
                    quad_matvec<G, RU, NXL, RX, NXP>(u, cK, x);    // Kinf x
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RX, NXL, RX, NXP>(xn, cA, x);   // A x (+ fdyn)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        u[m] = -u[m] - (RT)d_c[m];
                        uf[m] = (float)u[m];
                        float zn = uf[m] + y_c[m];
                        zn = fminf(bk[2 * RX + RU + m], fmaxf(bk[2 * RX + m], zn));
                        SU(Sy, k, m) = (y_c[m] + uf[m]) - zn;
                        pri_u = fmaxf(pri_u, fabsf(uf[m] - zn));
                        dua_u = fmaxf(dua_u, fabsf(z_c[m] - zn));
                        SU(Szw, k, m) = zn;
                    }
                    if constexpr (EXT) {
                        if (soc_u) {
                            float zc2[RU];
#pragma unroll
                            for (int m = 0; m < RU; ++m) zc2[m] = uf[m] + yc_c[m];
                            for (int c = 0; c < ncu; ++c) {
                                const uint4 mk = cm[c * G];
                                project_soc_quad<RU>(zc2, mk.z, mk.w, P.cu[c]);
                            }
#pragma unroll
                            for (int m = 0; m < RU; ++m) {
                                SU(Syc, k, m) = (yc_c[m] + uf[m]) - zc2[m];
                                pri_u = fmaxf(pri_u, fabsf(uf[m] - zc2[m]));
                                dua_u = fmaxf(dua_u, fabsf(zc_c[m] - zc2[m]));
                                SU(Szwc, k, m) = zc2[m];
                            }
                        }
                    }
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RX, NUL, RU, NUP>(xn, cB, u);   // + B u
#pragma unroll
                    for (int m = 0; m < RX; ++m) x[m] = xn[m];
                }
            }
            it += 1;
            if (need_res) {
                res0 = group_max<G>(pri_x);
                res1 = group_max<G>(dua_x) * rho;
                res2 = group_max<G>(pri_u);
                res3 = group_max<G>(dua_u) * rho;
                if (res0 < P.abs_pri_tol && res2 < P.abs_pri_tol && res1 < P.abs_dua_tol && res3 < P.abs_dua_tol)
                    conv = 1;
            }
            if (!conv) {
                // ================= fused backward sweep (admm.cpp:75-83, :196-197, :13-20) =================
                RT p[RX];
                {
                    RT acc[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) acc[m] = (RT)0;
                    if (P.ref_mode != REF_ZERO) {
                        RT xrl[RX];
#pragma unroll
                        for (int m = 0; m < RX; ++m) xrl[m] = (RT)ref_x(N - 1, m);
                        quad_matvec<G, RX, NXL, RX, NXP>(acc, cPT, xrl);
                    }
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        const float wN = SX(Sw, N - 1, m);
                        float tail = rho * (wN - SX(Sg, N - 1, m));
                        SX(Sv, N - 1, m) = wN;
                        if (soc_x) {
                            const float wc = SX(Swc, N - 1, m);
                            tail += rho * (wc - SX(Sgc, N - 1, m));
                            SX(Svc, N - 1, m) = wc;
                        }
                        p[m] = -acc[m] - (RT)tail;
                    }
                }
                float w_n[RX], g_n2[RX], wc_n[RX], gc_n2[RX], zw_n[RU], y_n2[RU], zwc_n[RU], yc_n2[RU];
                auto fetchb = [&](int k_) {
                    const int k = knot_sgpr(k_);
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        w_n[m] = SX(Sw, k, m);
                        g_n2[m] = SX(Sg, k, m);
                        wc_n[m] = soc_x ? SX(Swc, k, m) : 0.f;
                        gc_n2[m] = soc_x ? SX(Sgc, k, m) : 0.f;
                    }
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        zw_n[m] = SU(Szw, k, m);
                        y_n2[m] = SU(Sy, k, m);
                        zwc_n[m] = soc_u ? SU(Szwc, k, m) : 0.f;
                        yc_n2[m] = soc_u ? SU(Syc, k, m) : 0.f;
                    }
                };
                if (N >= 2) fetchb(N - 2);
                for (int k_ = N - 2; k_ >= 0; --k_) {
                    asm volatile("" ::: "memory");
                    const int k = knot_sgpr(k_);
                    RT r[RU], qk[RX];
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        float rr = -(ref_u(k, m) * cRD[m]);
                        rr -= rho * (zw_n[m] - y_n2[m]);
                        SU(Sz, k, m) = zw_n[m];
                        if (soc_u) {
                            rr -= rho * (zwc_n[m] - yc_n2[m]);
                            SU(Szc, k, m) = zwc_n[m];
                        }
                        r[m] = (RT)rr;
                    }
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        float qq = -(ref_x(k, m) * cQD[m]);
                        qq -= rho * (w_n[m] - g_n2[m]);
                        SX(Sv, k, m) = w_n[m];
                        if (soc_x) {
                            qq -= rho * (wc_n[m] - gc_n2[m]);
                            SX(Svc, k, m) = wc_n[m];
                        }
                        qk[m] = (RT)qq;
                    }
                    if (k > 0) fetchb(k - 1);
                    RT t[RU], dn[RU], ap[RX], kr[RX];
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        t[m] = r[m] + (EXT ? (RT)cBPF[m] : (RT)0);
                        dn[m] = (RT)0;
                    }
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RU, NXL, RX, NXP>(t, cBT, p);    // B^T p + r (+ BPf)
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RU, NUL, RU, NUP>(dn, cQI, t);   // d = Quu_inv (...)
#pragma unroll
                    for (int m = 0; m < RU; ++m) SU(Sd, k, m) = (float)dn[m];
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        ap[m] = qk[m] + (EXT ? (RT)cAPF[m] : (RT)0);
                        kr[m] = (RT)0;
                    }
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RX, NXL, RX, NXP>(ap, cAT, p);   // q + AmBKt p (+ APf)
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RX, NUL, RU, NUP>(kr, cKT, r);   // Kinf^T r
#pragma unroll
                    for (int m = 0; m < RX; ++m) p[m] = ap[m] - kr[m];
                }
            }
        }
        if (!__builtin_amdgcn_ballot_w64(active && !conv)) break;
    }

    if (active) {
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int m = 0; m < RX; ++m)
                if (q * RX + m < NX) P.xout[b * EX + k * NX + q * RX + m] = SX(Sw, k, m);
        for (int k = 0; k < N - 1; ++k)
#pragma unroll
            for (int m = 0; m < RU; ++m)
                if (q * RU + m < NU) P.uout[b * EU + k * NU + q * RU + m] = SU(Szw, k, m);
        if (q == 0) {
            P.iter[b] = it;
            P.solved[b] = conv;
            P.res[b * 4 + 0] = res0;
            P.res[b * 4 + 1] = res1;
            P.res[b * 4 + 2] = res2;
            P.res[b * 4 + 3] = res3;
        }
        if (P.save_state) {
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    const int row = q * RX + m;
                    if (row < NX) {
                        P.sg[b * EX + k * NX + row] = SX(Sg, k, m);
                        P.sv[b * EX + k * NX + row] = SX(Sv, k, m);
                        if (soc_x) {
                            P.sgc[b * EX + k * NX + row] = SX(Sgc, k, m);
                            P.svc[b * EX + k * NX + row] = SX(Svc, k, m);
                        }
                    }
                }
            for (int k = 0; k < N - 1; ++k)
#pragma unroll
                for (int m = 0; m < RU; ++m) {
                    const int row = q * RU + m;
                    if (row < NU) {
                        P.sy[b * EU + k * NU + row] = SU(Sy, k, m);
                        P.sz[b * EU + k * NU + row] = SU(Sz, k, m);
                        P.sd[b * EU + k * NU + row] = SU(Sd, k, m);
                        if (soc_u) {
                            P.syc[b * EU + k * NU + row] = SU(Syc, k, m);
                            P.szc[b * EU + k * NU + row] = SU(Szc, k, m);
                        }
                    }
                }
        }
    }
#undef SX
#undef SU
    {
        float m0 = active ? res0 : 0.f, m1 = active ? res1 : 0.f, m2 = active ? res2 : 0.f, m3 = active ? res3 : 0.f;
#pragma unroll
        for (int off = G; off < 64; off <<= 1) {
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
