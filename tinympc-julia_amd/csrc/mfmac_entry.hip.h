// Host-side pack builders + launcher for one (nx, nu) instantiation of the LDS-resident matrix-core kernel
// (admm_mfmac.hip.h).
#pragma once
#include <cstdlib>
#include <cstring>
#include <limits>

#include "admm_mfmac.hip.h"
#include "solver.h"

namespace tmpc {

// rows (of the stacked [x; u] vector) that lie in an enabled cone — the kernel derives the same count from AdmmParams
template <int NX, int NU>
int mfmac_cone_rows(const Solver &sv) {
    int n = 0;
    const int ncx = sv.st.en_state_soc ? sv.ncx : 0, ncu = sv.st.en_input_soc ? sv.ncu : 0;
    for (int r = 0; r < NX; ++r) {
        bool in = false;
        for (int c = 0; c < ncx; ++c) in = in || (r >= sv.Acx[c] && r < sv.Acx[c] + sv.qcx[c]);
        n += in;
    }
    for (int a = 0; a < NU; ++a) {
        bool in = false;
        for (int c = 0; c < ncu; ++c) in = in || (a >= sv.Acu[c] && a < sv.Acu[c] + sv.qcu[c]);
        n += in;
    }
    return n;
}

// do the enabled bounds depend on the knot?
inline bool mfmac_bounds_vary(const Solver &sv) { return sv.bounds_vary_by_knot(); }

// operand doubles of every lane, [field][64], then Pinf row-major [NX][NX].  For the 16 x 4 A operand of K-slice s lane l
// supplies tile row l % 16, tile column 4 s + l / 16; tile index t stands for x_t (t < 8) or u_{t-8} (8 <= t < 12).
template <int NX, int NU>
void build_mfmac_coef(const Solver &sv, std::vector<unsigned char> &out) {
    using S = ConeShape<NX, NU>;
    out.assign(((size_t)S::NF * 64 + (size_t)NX * NX) * sizeof(double), 0);
    double *o = reinterpret_cast<double *>(out.data());
    const Cache &c = sv.cache;
    auto is_x = [](int t) { return t < 8 && t < NX; };
    auto is_u = [](int t) { return t >= 8 && t - 8 < NU; };
    double Pf[NX], APf[NX], BPf[NU];
    for (int i = 0; i < NX; ++i) {
        Pf[i] = 0.0;
        for (int k = 0; k < NX; ++k) Pf[i] += c.Pinf(i, k) * sv.fdyn[k];
    }
    for (int i = 0; i < NX; ++i) {
        APf[i] = 0.0;
        for (int k = 0; k < NX; ++k) APf[i] += c.AmBKt(i, k) * Pf[k];
    }
    for (int a = 0; a < NU; ++a) {
        BPf[a] = 0.0;
        for (int k = 0; k < NX; ++k) BPf[a] += sv.B(k, a) * Pf[k];
    }
    for (int l = 0; l < 64; ++l) {
        const int m = l % 16, kq = l / 16;
        for (int s = 0; s < 3; ++s) {
            const int kc = 4 * s + kq;
            double mf = 0.0, mb = 0.0;
            if (is_x(m) && is_x(kc)) {
                mf = sv.A(m, kc);                       // (A - B Kinf)[m][kc], from A, B, Kinf themselves (set_cache_terms may
                for (int a = 0; a < NU; ++a) mf -= sv.B(m, a) * c.Kinf(a, kc);   // hand in an AmBKt that differs)
                mb = c.AmBKt(m, kc);
            } else if (is_x(m) && is_u(kc)) {
                for (int a = 0; a < NU; ++a) mf -= sv.B(m, a) * c.Quu_inv(a, kc - 8);   // - B Quu_inv t
                mb = -c.Kinf(kc - 8, m);                // - Kinf^T r
            } else if (is_u(m) && is_x(kc)) {
                mf = -c.Kinf(m - 8, kc);                // u = -Kinf x ...
                mb = sv.B(kc, m - 8);                   // B^T p (+ r through the accumulator's start value)
            } else if (is_u(m) && is_u(kc)) {
                mf = -c.Quu_inv(m - 8, kc - 8);         // ... - Quu_inv t
            } else if (kc == 11 && NU <= 3 && sv.has_fdyn) {
                // K index 11 carries the constant 1: the affine terms f | AmBKt Pinf f, B' Pinf f
                if (is_x(m)) mf = sv.fdyn[m], mb = APf[m];
                else if (is_u(m)) mb = BPf[m - 8];
            }
            o[(S::F_MF0 + s) * 64 + l] = mf;
            o[(S::F_MB0 + s) * 64 + l] = mb;
        }
        // per-lane constants of the lane's slots (state lane roles: group g = l / 16)
        const int g = l / 16;
        o[S::F_FD0 * 64 + l] = (sv.has_fdyn && g < NX) ? sv.fdyn[g] : 0.0;
        o[S::F_FD1 * 64 + l] = (sv.has_fdyn && 4 + g < NX) ? sv.fdyn[4 + g] : 0.0;
        o[S::F_APF0 * 64 + l] = (sv.has_fdyn && g < NX) ? APf[g] : 0.0;
        o[S::F_APF1 * 64 + l] = (sv.has_fdyn && 4 + g < NX) ? APf[4 + g] : 0.0;
        o[S::F_BPF * 64 + l] = (sv.has_fdyn && g < NU) ? BPf[g] : 0.0;
    }
    double *P = o + (size_t)S::NF * 64;
    for (int i = 0; i < NX; ++i)
        for (int k = 0; k < NX; ++k) P[i * NX + k] = c.Pinf(i, k);
}

template <int NX, int NU>
void build_mfmac_bounds(const Solver &sv, std::vector<float> &out) {
    using S = ConeShape<NX, NU>;
    constexpr float kInf = std::numeric_limits<float>::infinity();
    const int N = sv.N, nk = mfmac_bounds_vary(sv) ? N : 1;
    out.assign((size_t)S::bounds_len(nk), 0.f);
    for (int k = 0; k < nk; ++k) {
        float *p = out.data() + (size_t)k * 2 * S::NROW;
        for (int r = 0; r < NX; ++r) {
            p[r] = sv.st.en_state_bound ? (float)sv.x_min[r + (size_t)k * NX] : -kInf;
            p[S::NROW + r] = sv.st.en_state_bound ? (float)sv.x_max[r + (size_t)k * NX] : kInf;
        }
        for (int a = 0; a < NU; ++a) {
            const bool on = sv.st.en_input_bound && k < N - 1;
            p[NX + a] = on ? (float)sv.u_min[a + (size_t)k * NU] : -kInf;
            p[S::NROW + NX + a] = on ? (float)sv.u_max[a + (size_t)k * NU] : kInf;
        }
    }
    float *d = out.data() + (size_t)2 * S::NROW * nk;
    for (int r = 0; r < NX; ++r) d[r] = (float)sv.cache.Qd[r];
    for (int a = 0; a < NU; ++a) d[NX + a] = (float)sv.cache.Rd[a];
    d[S::NROW] = -kInf;
    d[S::NROW + 1] = kInf;
}

template <int NX, int NU>
size_t mfmac_lds_bytes(const Solver &sv) {
    const int mlx = sv.st.en_state_linear ? sv.mlx : 0, mlu = sv.st.en_input_linear ? sv.mlu : 0;
    const int lin_rows = (mlx > 0 ? NX : 0) + (mlu > 0 ? NU : 0);
    return ConeShape<NX, NU>::lds_bytes(sv.N, mfmac_bounds_vary(sv) ? sv.N : 1, mfmac_cone_rows<NX, NU>(sv) + lin_rows, mlx, mlu);
}

template <int NX, int NU>
size_t mfmac_scratch_floats(const Solver &sv) {
    return ((size_t)sv.batch + 15) / 16 * ConeShape<NX, NU>::scratch_floats(sv.N);
}

template <int NX, int NU>
hipError_t launch_mfmac(const AdmmParams &P_, bool ext, size_t lds, hipStream_t stream) {
    AdmmParams P = P_;
#ifdef TMPC_MFMAC_PROBE
    P.mpc_steps = std::getenv("TINYMPC_HIP_MFMAC_DEBUG") ? std::atoi(std::getenv("TINYMPC_HIP_MFMAC_DEBUG")) : 0;   // timing probe build only
#endif
    const int tiles = (P.batch + 15) / 16;
    // persistent workgroups (the kernel takes tiles off a counter): as many as fit on the chip at once
    const int cus = device_cu_count();   // (per device: a sharded handle launches on several)
#define TMPC_MFMAC_LAUNCH(REFS_, CX_, CU_, BV_, LIN_)                                                                     \
    do {                                                                                                              \
        (void)hipFuncSetAttribute((const void *)admm_mfmac_kernel<NX, NU, REFS_, CX_, CU_, BV_, LIN_>,                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        int per_cu = 0;                                                                                               \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, admm_mfmac_kernel<NX, NU, REFS_, CX_, CU_, BV_, LIN_>,    \
                                                         192, lds) != hipSuccess || per_cu <= 0)                      \
            per_cu = 1;                                                                                               \
        const int grid = tiles < per_cu * cus ? tiles : per_cu * cus;                                                 \
        hipLaunchKernelGGL((admm_mfmac_kernel<NX, NU, REFS_, CX_, CU_, BV_, LIN_>), dim3(grid), dim3(192), lds, stream, P);  \
    } while (0)
    // (the kernel template also has two cones on a side — CX / CU = 2 — and linear rows — LIN — from round 3: correct, but no
    // faster than the stream kernel, and since round 4 those layouts run on the transposed-sets kernel specialised for them
    // (jit.cpp); they are not instantiated any more)
#define TMPC_MFMAC_LAUNCH_BV(REFS_, CX_, CU_)                                                                         \
    do {                                                                                                              \
        if (P.bounds_stride) TMPC_MFMAC_LAUNCH(REFS_, CX_, CU_, true, false); else TMPC_MFMAC_LAUNCH(REFS_, CX_, CU_, false, false); \
    } while (0)
#define TMPC_MFMAC_LAUNCH_CU(REFS_, CX_)                                                    \
    do {                                                                                   \
        if (P.ncu > 0) TMPC_MFMAC_LAUNCH_BV(REFS_, CX_, 1);                                \
        else TMPC_MFMAC_LAUNCH_BV(REFS_, CX_, 0);                                          \
    } while (0)
#define TMPC_MFMAC_LAUNCH_C(REFS_)                                                         \
    do {                                                                                   \
        if (P.ncx > 0) TMPC_MFMAC_LAUNCH_CU(REFS_, 1);                                     \
        else TMPC_MFMAC_LAUNCH_CU(REFS_, 0);                                               \
    } while (0)
    if (P.ncx > 1 || P.ncu > 1 || P.mlx + P.mlu > 0) return hipErrorInvalidValue;   // (the routes do not send these here)
    (void)ext;
    if (P.ref_mode == REF_ZERO) TMPC_MFMAC_LAUNCH_C(REF_ZERO); else TMPC_MFMAC_LAUNCH_C(REF_SHARED);
#undef TMPC_MFMAC_LAUNCH_C
#undef TMPC_MFMAC_LAUNCH_CU
#undef TMPC_MFMAC_LAUNCH_BV
#undef TMPC_MFMAC_LAUNCH
    return hipGetLastError();
}

#define TMPC_DEFINE_MFMAC_ENTRY(NX, NU)                                                                              \
    const ConeEntry *mfmac_entry_##NX##_##NU() {                                                                    \
        static const ConeEntry e = {NX, NU, 0, nullptr, false, "mfmac<" #NX "," #NU ">", &build_mfmac_coef<NX, NU>,                    \
                                    &build_mfmac_bounds<NX, NU>, &mfmac_lds_bytes<NX, NU>, &mfmac_scratch_floats<NX, NU>, \
                                    &mfmac_bounds_vary, &launch_mfmac<NX, NU>};                                     \
        return &e;                                                                                                  \
    }

}  // namespace tmpc
