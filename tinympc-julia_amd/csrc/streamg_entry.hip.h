// Host-side pack builders + launcher for one (nx, nu, G) instantiation of the stream kernel (admm_streamg.hip.h).
#pragma once
#include <cstring>
#include <limits>

#include "admm_streamg.hip.h"
#include "solver.h"

namespace tmpc {

// One family's rows for the G lane roles; put(q, idx, value) stores element idx of role q's pack.
// A, B column-major (nx x nx, nx x nu) as the reference passes them (bindings.cpp:24-31).
template <int NX, int NU, int G, class Put>
void fill_streamg_family(const double *A, const double *Bm, const Cache &c, const std::vector<double> &fdyn,
                         Put put) {
    using PK = StreamPackG<NX, NU, G>;
    using S = typename PK::S;
    constexpr int RX = S::RX, RU = S::RU, NXP = S::NXP, NUP = S::NUP;
    double Pf[NX];
    for (int i = 0; i < NX; ++i) {
        Pf[i] = 0.0;
        for (int l = 0; l < NX; ++l) Pf[i] += c.Pinf(i, l) * fdyn[l];
    }
    for (int q = 0; q < G; ++q) {
        for (int m = 0; m < RX; ++m) {
            const int row = q * RX + m;
            if (row >= NX) continue;
            double apf = 0.0;
            for (int j = 0; j < NX; ++j) {
                put(q, S::O_A + m * NXP + j, A[row + (size_t)j * NX]);
                put(q, S::O_AT + m * NXP + j, c.AmBKt(row, j));
                put(q, S::O_PT + m * NXP + j, c.Pinf(j, row));
                put(q, PK::O_ATT + m * NXP + j, A[j + (size_t)row * NX]);   // A^T (adaptive rho: A' g)
                apf += c.AmBKt(row, j) * Pf[j];
            }
            for (int a = 0; a < NU; ++a) {
                put(q, S::O_B + m * NUP + a, Bm[row + (size_t)a * NX]);
                put(q, S::O_KT + m * NUP + a, c.Kinf(a, row));
            }
            put(q, PK::O_F + m, fdyn[row]);
            put(q, PK::O_APF + m, apf);
        }
        for (int m = 0; m < RU; ++m) {
            const int row = q * RU + m;  // no replication of a single input: lanes without a row carry zeros
            if (row >= NU) continue;
            double bpf = 0.0;
            for (int j = 0; j < NX; ++j) {
                put(q, S::O_K + m * NXP + j, c.Kinf(row, j));
                put(q, S::O_BT + m * NXP + j, Bm[j + (size_t)row * NX]);
                bpf += Bm[j + (size_t)row * NX] * Pf[j];
            }
            for (int a = 0; a < NU; ++a) put(q, S::O_QI + m * NUP + a, c.Quu_inv(row, a));
            put(q, PK::O_BPF + m, bpf);
        }
    }
}

template <int NX, int NU, int G, class RT>
void fill_streamg_coef(const Solver &sv, std::vector<unsigned char> &out) {
    using PK = StreamPackG<NX, NU, G>;
    if (!sv.hetero) {  // [role][CP]
        out.assign((size_t)G * PK::CP * sizeof(RT), 0);
        RT *o = reinterpret_cast<RT *>(out.data());
        std::vector<double> A((size_t)NX * NX), Bm((size_t)NX * NU);
        for (int j = 0; j < NX; ++j)
            for (int i = 0; i < NX; ++i) A[i + (size_t)j * NX] = sv.A(i, j);
        for (int a = 0; a < NU; ++a)
            for (int i = 0; i < NX; ++i) Bm[i + (size_t)a * NX] = sv.B(i, a);
        fill_streamg_family<NX, NU, G>(A.data(), Bm.data(), sv.cache, sv.fdyn,
                                    [&](int q, int idx, double v) { o[(size_t)q * PK::CP + idx] = (RT)v; });
        return;
    }
    // one family per instance: [element][G*batch + role], each lane reads its own column
    const size_t Bn = (size_t)sv.batch, B4 = G * Bn;
    out.assign((size_t)PK::CP * B4 * sizeof(RT), 0);
    RT *o = reinterpret_cast<RT *>(out.data());
    for (size_t b = 0; b < Bn; ++b)
        fill_streamg_family<NX, NU, G>(sv.het_A.data() + b * NX * NX, sv.het_B.data() + b * NX * NU, sv.het_cache[b],
                                    sv.fdyn, [&](int q, int idx, double v) { o[(size_t)idx * B4 + G * b + q] = (RT)v; });
}

template <int NX, int NU, int G>
void build_streamg_coef(const Solver &sv, std::vector<unsigned char> &out) {
    if (sv.precision == 0)
        fill_streamg_coef<NX, NU, G, double>(sv, out);
    else
        fill_streamg_coef<NX, NU, G, float>(sv, out);
}

// [N][role][xmin RX | xmax RX | umin RU | umax RU], then [role][diag(Q)+rho RX | diag(R)+rho RU]
template <int NX, int NU, int G>
void build_streamg_bounds(const Solver &sv, std::vector<float> &out) {
    using PK = StreamPackG<NX, NU, G>;
    constexpr float kInf = std::numeric_limits<float>::infinity();
    constexpr int RX = PK::RX, RU = PK::RU, BW = PK::BW, DW = PK::DW;
    const int N = sv.N;
    out.assign((size_t)N * G * BW + G * DW, 0.f);
    for (int k = 0; k < N; ++k)
        for (int q = 0; q < G; ++q) {
            float *p = out.data() + ((size_t)k * G + q) * BW;
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                const bool on = sv.st.en_state_bound && row < NX;
                p[m] = on ? (float)sv.x_min[row + (size_t)k * NX] : -kInf;
                p[RX + m] = on ? (float)sv.x_max[row + (size_t)k * NX] : kInf;
            }
            for (int m = 0; m < RU; ++m) {
                const int row = q * RU + m;
                const bool on = sv.st.en_input_bound && row < NU && k < N - 1;
                p[2 * RX + m] = on ? (float)sv.u_min[row + (size_t)k * NU] : -kInf;
                p[2 * RX + RU + m] = on ? (float)sv.u_max[row + (size_t)k * NU] : kInf;
            }
        }
    float *dg = out.data() + (size_t)N * G * BW;
    for (int q = 0; q < G; ++q) {
        for (int m = 0; m < RX; ++m) dg[q * DW + m] = q * RX + m < NX ? (float)sv.cache.Qd[q * RX + m] : 0.f;
        for (int m = 0; m < RU; ++m) dg[q * DW + RX + m] = q * RU + m < NU ? (float)sv.cache.Rd[q * RU + m] : 0.f;
    }
}

template <int NX, int NU, int G>
size_t streamg_lds_bytes(int N, int precision) {
    using PK = StreamPackG<NX, NU, G>;
    const size_t rt = precision == 0 ? 8 : 4;
    return rt * G * PK::CP + 4 * ((size_t)N * G * PK::BW + G * PK::DW);
}

// floats of scratch per instance: 3 state-shaped and 3 input-shaped arrays per constraint set (box | + cones |
// + linear inequalities), + d
template <int NX, int NU>
size_t streamg_scratch_floats(int N, int sets) {
    return (size_t)NX * N * (3 * sets) + (size_t)NU * (N - 1) * (3 * sets + 1);
}

template <int NX, int NU, int G>
hipError_t launch_streamg(const AdmmParams &P, int precision, int ext, bool het, hipStream_t stream) {
    const int grid = (P.batch + 256 / G - 1) / (256 / G);
    const size_t lds = streamg_lds_bytes<NX, NU, G>(P.N, precision);
    const bool oneshot = P.cold_start && !P.save_state;  // nothing of the workspace outlives the launch
#define TMPC_LAUNCH(RT_, EXT_, HET_, OS_)                                                                          \
    do {                                                                                                           \
        if (lds > 48 * 1024)                                                                                       \
            (void)hipFuncSetAttribute((const void *)admm_streamg_kernel<NX, NU, G, RT_, EXT_, HET_, OS_>,          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        hipLaunchKernelGGL((admm_streamg_kernel<NX, NU, G, RT_, EXT_, HET_, OS_>), dim3(grid), dim3(256), lds, stream, \
                           P);                                                                                     \
    } while (0)
#define TMPC_LAUNCH_OS(RT_, EXT_, HET_)                                                            \
    do {                                                                                           \
        if (oneshot) TMPC_LAUNCH(RT_, EXT_, HET_, true); else TMPC_LAUNCH(RT_, EXT_, HET_, false); \
    } while (0)
#define TMPC_LAUNCH_EXT(RT_, HET_)                       \
    do {                                                 \
        if (ext == 2) TMPC_LAUNCH_OS(RT_, 2, HET_);      \
        else if (ext == 1) TMPC_LAUNCH_OS(RT_, 1, HET_); \
        else TMPC_LAUNCH_OS(RT_, 0, HET_);               \
    } while (0)
#define TMPC_LAUNCH_RT(RT_)                                                        \
    do {                                                                           \
        if (het) TMPC_LAUNCH_EXT(RT_, true); else TMPC_LAUNCH_EXT(RT_, false);     \
    } while (0)
    if (P.adaptive_rho) {   // one family, box sets only (the solver checks): rho, Kinf, Pinf per instance
#define TMPC_LAUNCH_ADP(RT_, OS_)                                                                                  \
    do {                                                                                                           \
        if (lds > 48 * 1024)                                                                                       \
            (void)hipFuncSetAttribute((const void *)admm_streamg_kernel<NX, NU, G, RT_, 0, false, OS_, true>,      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        hipLaunchKernelGGL((admm_streamg_kernel<NX, NU, G, RT_, 0, false, OS_, true>), dim3(grid), dim3(256), lds, \
                           stream, P);                                                                             \
    } while (0)
        if (precision == 0) {
            if (oneshot) TMPC_LAUNCH_ADP(double, true); else TMPC_LAUNCH_ADP(double, false);
        } else {
            if (oneshot) TMPC_LAUNCH_ADP(float, true); else TMPC_LAUNCH_ADP(float, false);
        }
#undef TMPC_LAUNCH_ADP
    } else if (precision == 0) TMPC_LAUNCH_RT(double); else TMPC_LAUNCH_RT(float);
#undef TMPC_LAUNCH_RT
#undef TMPC_LAUNCH_EXT
#undef TMPC_LAUNCH_OS
#undef TMPC_LAUNCH
    return hipGetLastError();
}

#define TMPC_DEFINE_STREAMG_ENTRY(NX, NU, GG)                                                                       \
    const StreamEntry *stream##GG##_entry_##NX##_##NU() {                                                          \
        static const StreamEntry e = {NX, NU, GG, "stream" #GG "<" #NX "," #NU ">", &build_streamg_coef<NX, NU, GG>, \
                                      &build_streamg_bounds<NX, NU, GG>, &streamg_lds_bytes<NX, NU, GG>,           \
                                      &streamg_scratch_floats<NX, NU>, &launch_streamg<NX, NU, GG>};               \
        return &e;                                                                                                 \
    }

}  // namespace tmpc
