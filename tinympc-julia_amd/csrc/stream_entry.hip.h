// Host-side pack builders + launcher for one (nx, nu) instantiation of the stream kernel.
#pragma once
#include <cstring>
#include <limits>

#include "admm_stream.hip.h"
#include "solver.h"

namespace tmpc {

template <int NX, int NU, class RT>
void fill_stream_coef(const Solver &sv, std::vector<unsigned char> &out) {
    using PK = StreamPack<NX, NU>;
    out.assign((size_t)PK::LEN * sizeof(RT), 0);
    auto put = [&](size_t idx, double val) {
        const RT v = (RT)val;
        std::memcpy(out.data() + idx * sizeof(RT), &v, sizeof(RT));
    };
    const Cache &c = sv.cache;
    std::vector<double> Pf(NX, 0.0);
    for (int i = 0; i < NX; ++i)
        for (int l = 0; l < NX; ++l) Pf[i] += c.Pinf(i, l) * sv.fdyn[l];
    for (int r = 0; r < NX; ++r) {
        double apf = 0.0;
        for (int j = 0; j < NX; ++j) {
            put(PK::O_A + r * NX + j, sv.A(r, j));
            put(PK::O_AT + r * NX + j, c.AmBKt(r, j));
            put(PK::O_PT + r * NX + j, c.Pinf(j, r));  // (Pinf^T)[r][j]
            apf += c.AmBKt(r, j) * Pf[j];
        }
        for (int a = 0; a < NU; ++a) {
            put(PK::O_B + r * NU + a, sv.B(r, a));
            put(PK::O_KT + r * NU + a, c.Kinf(a, r));
        }
        put(PK::O_F + r, sv.fdyn[r]);
        put(PK::O_APF + r, apf);
    }
    for (int a = 0; a < NU; ++a) {
        double bpf = 0.0;
        for (int j = 0; j < NX; ++j) {
            put(PK::O_K + a * NX + j, c.Kinf(a, j));
            put(PK::O_BT + a * NX + j, sv.B(j, a));
            bpf += sv.B(j, a) * Pf[j];
        }
        for (int c2 = 0; c2 < NU; ++c2) put(PK::O_QI + a * NU + c2, c.Quu_inv(a, c2));
        put(PK::O_BPF + a, bpf);
    }
}

// One family per instance: the same pack, laid out [element][batch] so that each lane reads its own column.
template <int NX, int NU, class RT>
void fill_stream_coef_het(const Solver &sv, std::vector<unsigned char> &out) {
    using PK = StreamPack<NX, NU>;
    const size_t Bn = (size_t)sv.batch;
    out.assign((size_t)PK::LEN * Bn * sizeof(RT), 0);
    RT *o = reinterpret_cast<RT *>(out.data());
    for (size_t b = 0; b < Bn; ++b) {
        const double *A = sv.het_A.data() + b * NX * NX, *Bm = sv.het_B.data() + b * NX * NU;
        const Cache &c = sv.het_cache[b];
        auto put = [&](int idx, double val) { o[(size_t)idx * Bn + b] = (RT)val; };
        std::vector<double> Pf(NX, 0.0);
        for (int i = 0; i < NX; ++i)
            for (int l = 0; l < NX; ++l) Pf[i] += c.Pinf(i, l) * sv.fdyn[l];
        for (int r = 0; r < NX; ++r) {
            double apf = 0.0;
            for (int j = 0; j < NX; ++j) {
                put(PK::O_A + r * NX + j, A[r + (size_t)j * NX]);
                put(PK::O_AT + r * NX + j, c.AmBKt(r, j));
                put(PK::O_PT + r * NX + j, c.Pinf(j, r));
                apf += c.AmBKt(r, j) * Pf[j];
            }
            for (int a = 0; a < NU; ++a) {
                put(PK::O_B + r * NU + a, Bm[r + (size_t)a * NX]);
                put(PK::O_KT + r * NU + a, c.Kinf(a, r));
            }
            put(PK::O_F + r, sv.fdyn[r]);
            put(PK::O_APF + r, apf);
        }
        for (int a = 0; a < NU; ++a) {
            double bpf = 0.0;
            for (int j = 0; j < NX; ++j) {
                put(PK::O_K + a * NX + j, c.Kinf(a, j));
                put(PK::O_BT + a * NX + j, Bm[j + (size_t)a * NX]);
                bpf += Bm[j + (size_t)a * NX] * Pf[j];
            }
            for (int c2 = 0; c2 < NU; ++c2) put(PK::O_QI + a * NU + c2, c.Quu_inv(a, c2));
            put(PK::O_BPF + a, bpf);
        }
    }
}

template <int NX, int NU>
void build_stream_coef(const Solver &sv, std::vector<unsigned char> &out) {
    if (sv.hetero) {
        if (sv.precision == 0)
            fill_stream_coef_het<NX, NU, double>(sv, out);
        else
            fill_stream_coef_het<NX, NU, float>(sv, out);
        return;
    }
    if (sv.precision == 0)
        fill_stream_coef<NX, NU, double>(sv, out);
    else
        fill_stream_coef<NX, NU, float>(sv, out);
}

template <int NX, int NU>
void build_stream_bounds(const Solver &sv, std::vector<float> &out) {
    constexpr float kInf = std::numeric_limits<float>::infinity();
    const int N = sv.N, W = 2 * NX + 2 * NU;
    out.assign((size_t)N * W + NX + NU, 0.f);
    for (int k = 0; k < N; ++k) {
        float *p = out.data() + (size_t)k * W;
        for (int m = 0; m < NX; ++m) {
            p[m] = sv.st.en_state_bound ? (float)sv.x_min[m + (size_t)k * NX] : -kInf;
            p[NX + m] = sv.st.en_state_bound ? (float)sv.x_max[m + (size_t)k * NX] : kInf;
        }
        for (int a = 0; a < NU; ++a) {
            const bool on = sv.st.en_input_bound && k < N - 1;
            p[2 * NX + a] = on ? (float)sv.u_min[a + (size_t)k * NU] : -kInf;
            p[2 * NX + NU + a] = on ? (float)sv.u_max[a + (size_t)k * NU] : kInf;
        }
    }
    for (int m = 0; m < NX; ++m) out[(size_t)N * W + m] = (float)sv.cache.Qd[m];
    for (int a = 0; a < NU; ++a) out[(size_t)N * W + NX + a] = (float)sv.cache.Rd[a];
}

template <int NX, int NU>
size_t stream_lds_bytes(int N, int precision) {
    const size_t rt = precision == 0 ? 8 : 4;
    return rt * ((StreamPack<NX, NU>::LEN + 1) & ~1) + 4 * ((size_t)N * (2 * NX + 2 * NU) + NX + NU);
}

template <int NX, int NU>
size_t stream_scratch_floats(int N, bool cones) {
    const size_t EX = (size_t)NX * N, EU = (size_t)NU * (N - 1);
    return cones ? 6 * EX + 7 * EU : 3 * EX + 4 * EU;
}

template <int NX, int NU>
hipError_t launch_stream(const AdmmParams &P, int precision, bool ext, bool het, hipStream_t stream) {
    const int grid = (P.batch + 255) / 256;
    const size_t lds = stream_lds_bytes<NX, NU>(P.N, precision);
#define TMPC_LAUNCH(RT_, EXT_, HET_)                                                                      \
    do {                                                                                                  \
        if (lds > 48 * 1024)                                                                              \
            (void)hipFuncSetAttribute((const void *)admm_stream_kernel<NX, NU, RT_, EXT_, HET_>,          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
        hipLaunchKernelGGL((admm_stream_kernel<NX, NU, RT_, EXT_, HET_>), dim3(grid), dim3(256), lds, stream, \
                           P);                                                                            \
    } while (0)
#define TMPC_LAUNCH_RT(RT_)                                                \
    do {                                                                   \
        if (het) {                                                         \
            if (ext) TMPC_LAUNCH(RT_, true, true); else TMPC_LAUNCH(RT_, false, true);   \
        } else {                                                           \
            if (ext) TMPC_LAUNCH(RT_, true, false); else TMPC_LAUNCH(RT_, false, false); \
        }                                                                  \
    } while (0)
    if (precision == 0) TMPC_LAUNCH_RT(double); else TMPC_LAUNCH_RT(float);
#undef TMPC_LAUNCH_RT
#undef TMPC_LAUNCH
    return hipGetLastError();
}

#define TMPC_DEFINE_STREAM_ENTRY(NX, NU)                                                             \
    const StreamEntry *stream_entry_##NX##_##NU() {                                                  \
        static const StreamEntry e = {NX, NU, 1, "stream<" #NX "," #NU ">", &build_stream_coef<NX, NU>, \
                                      &build_stream_bounds<NX, NU>, &stream_lds_bytes<NX, NU>,       \
                                      &stream_scratch_floats<NX, NU>, &launch_stream<NX, NU>};       \
        return &e;                                                                                   \
    }

}  // namespace tmpc
