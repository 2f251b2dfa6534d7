// Fused ADMM kernel with a run-time horizon, G = 1, 2 or 4 lanes per instance: "stream<G>".
//
// For shapes / options without an unrolled quad-kernel instantiation (admm_quad.hip.h): any horizon, the
// affine dynamics term, second-order cones, linear inequalities, one problem family per instance.  The knot loops are rolled and the
// per-instance trajectories are streamed through a scratch block in HBM / Infinity Cache, D knots ahead of
// their use; the lane mapping is the quad kernel's (lane q of a group owns state rows [q*RX,(q+1)*RX) and
// input rows [q*RU,(q+1)*RU), mat-vec operands fetched with DPP quad_perm broadcasts).
//   * G trades instruction efficiency against parallelism: one lane per instance executes the fewest
//     instructions per instance but fills the chip only from ~65 536 instances up; four lanes put 4x the
//     wavefronts in flight at ~2x the instructions.  G = 4 is what is instantiated (kernels.hip): measured, 1 and 2
//     lanes are no faster at any batch size, because from ~32 768 instances up all three are bound by the state stream.
//   * scratch layout [array][knot][instance][row] with the real nx / nu rows only: a lane's rows are one
//     contiguous (vector) access, a wavefront's instances one contiguous span, no padding rows travel;
//   * one-shot solves (cold start, workspace not kept: OS) update vnew / znew in place and hand the backward
//     sweep one fused array (vnew - g [+ the cone set's]) instead of four: 10 instead of 14 state-shaped and
//     12 instead of 16 input-shaped float transfers per knot and iteration (6 and 8 when no residual check
//     reads vnew / znew back);
//   * one family for the batch: coefficient rows per lane role in LDS (conflict-free image, as in the
//     quad kernel); one family per instance (HET): the same rows as per-lane columns in HBM;
//   * second-order cones may straddle lanes: squared head norms and the axis value are summed over
//     the group with DPP steps.
#pragma once
#include <hip/hip_runtime.h>

#include "admm_params.h"
#include "admm_quad.hip.h"

namespace tmpc {

// Pack per lane role: the quad kernel's three blocks (QuadShape offsets; N plays no role in them) + the affine block
template <int NX, int NU, int G>
struct StreamPackG {
    using S = QuadShape<NX, NU, 2, G>;
    static constexpr int RX = S::RX, RU = S::RU;
    static constexpr int O_F = S::CP, O_APF = O_F + RX, O_BPF = O_APF + RX;
    static constexpr int O_ATT = S::pad8(O_BPF + RU);   // A^T rows [RX][NXP] (adaptive rho: the dual residual's A' g)
    static constexpr int CP = O_ATT + S::pad8(RX * S::NXP);
    // adaptive rho (ADP): the rows that differ per instance, as this lane's column of a [ADP_LEN][G*batch] matrix
    static constexpr int AO_K = 0, AO_KT = AO_K + RU * S::NXP, AO_PT = AO_KT + RX * S::NUP, ADP_LEN = AO_PT + RX * S::NXP;
    static constexpr int BW = 2 * RX + 2 * RU;  // bounds per knot and role: xmin xmax umin umax
    static constexpr int DW = RX + RU;          // diag(Q)+rho, diag(R)+rho per role
};

// the (wave-uniform) knot index, hidden from loop strength reduction
__device__ __forceinline__ int knot_sgpr(int k) {
    k = __builtin_amdgcn_readfirstlane(k);  // free when the compiler already knows k to be uniform; makes "s" legal when not
    asm volatile("" : "+s"(k));
    return k;
}

// a wave-uniform pointer pinned to SGPRs: accesses become (SGPR base) + (32-bit VGPR offset)
template <class T>
__device__ __forceinline__ T *sgpr_ptr(T *p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    p = reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
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

// R consecutive floats of one lane as the widest accesses (4-byte aligned; the hardware takes unaligned vectors)
typedef float sg_f2 __attribute__((ext_vector_type(2)));
typedef float sg_f4 __attribute__((ext_vector_type(4)));
typedef sg_f2 sg_f2u __attribute__((aligned(4)));
typedef sg_f4 sg_f4u __attribute__((aligned(4)));
template <int R, int O = 0, int RT_>
__device__ __forceinline__ void load_rows(const float __attribute__((address_space(1))) * p, float (&dst)[RT_]) {
    if constexpr (R >= 4) {
        const sg_f4 v = *(const sg_f4u __attribute__((address_space(1))) *)(p + O);
        dst[O] = v.x, dst[O + 1] = v.y, dst[O + 2] = v.z, dst[O + 3] = v.w;
        load_rows<R - 4, O + 4>(p, dst);
    } else if constexpr (R >= 2) {
        const sg_f2 v = *(const sg_f2u __attribute__((address_space(1))) *)(p + O);
        dst[O] = v.x, dst[O + 1] = v.y;
        load_rows<R - 2, O + 2>(p, dst);
    } else if constexpr (R == 1) {
        dst[O] = p[O];
    }
}
template <int R, int O = 0, int RT_>
__device__ __forceinline__ void store_rows(float __attribute__((address_space(1))) * p, const float (&src)[RT_]) {
    if constexpr (R >= 4) {
        sg_f4 v;
        v.x = src[O], v.y = src[O + 1], v.z = src[O + 2], v.w = src[O + 3];
        *(sg_f4u __attribute__((address_space(1))) *)(p + O) = v;
        store_rows<R - 4, O + 4>(p, src);
    } else if constexpr (R >= 2) {
        sg_f2 v;
        v.x = src[O], v.y = src[O + 1];
        *(sg_f2u __attribute__((address_space(1))) *)(p + O) = v;
        store_rows<R - 2, O + 2>(p, src);
    } else if constexpr (R == 1) {
        p[O] = src[O];
    }
}

// Coefficient rows of one lane role in the LDS image (CoefLds's role-interleaved 16-byte chunks), any offset
template <class RT, int G>
struct CoefRole {
    static constexpr int VEC = 16 / (int)sizeof(RT);
    const RT *base;  // s_coef + q * VEC
    int off;
    __device__ __forceinline__ RT operator[](int i) const {
        const int j = off + i;
        return base[(j / VEC) * G * VEC + j % VEC];
    }
    __device__ __forceinline__ CoefRole operator+(int o) const { return CoefRole{base, off + o}; }
};

// One family per instance: element i of this lane's pack is column `lane` of row i of a [CP][G*batch] matrix.
template <class RT>
struct CoefCol {
    const RT *base;  // wave-uniform
    long stride;
    unsigned lane;  // byte offset of this lane's column
    __device__ __forceinline__ RT operator[](int i) const { return *lane_elem(sgpr_ptr(base + i * stride), lane); }
    __device__ __forceinline__ CoefCol operator+(int off) const { return CoefCol{base + off * stride, stride, lane}; }
};

template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (G >= 2) v += dpp_quad<0xB1>(v);
    if constexpr (G == 4) v += dpp_quad<0x4E>(v);
    return v;
}

// max over the G lanes of a group, any arithmetic type (the adaptive-rho norms are kept in the kernel's RT)
template <int G, class T>
__device__ __forceinline__ T group_max_t(T v) {
    if constexpr (G >= 2) {
        const T o = __shfl_xor(v, 1, 64);
        v = o > v ? o : v;
    }
    if constexpr (G == 4) {
        const T o = __shfl_xor(v, 2, 64);
        v = o > v ? o : v;
    }
    return v;
}
template <class T>
__device__ __forceinline__ void upmax_abs(T &m, T v) {
    v = v < (T)0 ? -v : v;
    m = v > m ? v : m;
}

// cone c restricted to this lane's R local rows: bit m of head / axis set when local row m belongs to it
template <int G, int R>
__device__ __forceinline__ void project_soc_group(float (&blk)[R], unsigned head, unsigned axis, float mu) {
    float a2 = 0.f, ax = 0.f;
#pragma unroll
    for (int m = 0; m < R; ++m) {
        if ((head >> m) & 1u) a2 = fmaf(blk[m], blk[m], a2);
        if ((axis >> m) & 1u) ax = blk[m];
    }
    a2 = group_sum<G>(a2);
    ax = group_sum<G>(ax);  // exactly one lane contributes
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

// z <- projection onto {a_k . z <= b_k}, one row after the other; a lane holds its R local entries of each row
// (rows + k * stride), the dot product is summed over the group, so the branch is uniform within it
template <int G, int R>
__device__ __forceinline__ void project_halfspaces_group(float (&z)[R], const float *rows, int stride, int m,
                                                         const float *b, const float *n2) {
    for (int k = 0; k < m; ++k) {
        const float *a = rows + k * stride;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j) dot = fmaf(a[j], z[j], dot);
        dot = group_sum<G>(dot);
        if (dot > b[k]) {
            const float t = (dot - b[k]) / n2[k];
#pragma unroll
            for (int j = 0; j < R; ++j) z[j] -= t * a[j];
        }
    }
}

// wavefronts per SIMD the register allocation is held to, and knots of prefetch, per group size (MI355X, rocket
// N=50 with cones: 2..4 wavefronts and depth 1..3 are within 5 % of each other for G = 4; scripts/stream_tune.sh)
template <int G>
struct StreamTune {
    static constexpr int WAVES = G == 4 ? 3 : (G == 2 ? 2 : 1);
    static constexpr int DEPTH = G == 4 ? 1 : (G == 2 ? 2 : 3);
};
#ifndef TMPC_STREAM_DEPTH
#define TMPC_STREAM_DEPTH(G) StreamTune<G>::DEPTH
#endif
#ifndef TMPC_STREAM_WAVES
#define TMPC_STREAM_WAVES(G) StreamTune<G>::WAVES
#endif

// ADP: adaptive rho (admm.cpp:147-174 with rho_benchmark.cpp:44-213).  One family for the batch (its rows in LDS), but
// rho, Kinf and Pinf are every instance's own: the rows built from them (Kinf, Kinf^T, Pinf^T) are read from the lane's
// column of a scratch matrix that the kernel fills from the solver's adaptive state at entry and re-writes, together
// with that state, whenever it adapts — every 5th iteration, from norms gathered during that iteration's forward sweep.
template <int NX, int NU, int G, class RT, int EXT, bool HET, bool OS, bool ADP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TMPC_STREAM_WAVES(G)))) void admm_streamg_kernel(const AdmmParams P) {
    static_assert(!ADP || (!HET && EXT == 0), "adaptive rho: one family, box sets only");
    using PK = StreamPackG<NX, NU, G>;
    using S = typename PK::S;
    constexpr int T = 256, D = TMPC_STREAM_DEPTH(G), RX = S::RX, RU = S::RU, NXP = S::NXP, NUP = S::NUP, NXL = S::NXL,
                  NUL = S::NUL;
    constexpr bool XFULL = NX % RX == 0, UFULL = NU % RU == 0;  // every owning lane owns RX / RU real rows
    constexpr bool ALLX = XFULL && NXL == G, ALLU = UFULL && NUL == G;  // ... and every lane is an owning lane
    extern __shared__ __align__(16) unsigned char s_rawg[];
    RT *s_coef = reinterpret_cast<RT *>(s_rawg);
    float *s_bnd = reinterpret_cast<float *>(s_rawg + sizeof(RT) * G * PK::CP);
    __shared__ uint4 s_cmask[8 * G];
    // linear inequalities (EXT == 2): per row and lane role the local entries, then right-hand sides and |a|^2
    __shared__ float s_lax[EXT == 2 ? LIN_MAX_ROWS * G * RX : 1], s_lau[EXT == 2 ? LIN_MAX_ROWS * G * RU : 1],
        s_lb[4 * LIN_MAX_ROWS];

    const int N = P.N;
    const int tid = threadIdx.x;
    const RT *gcoef = reinterpret_cast<const RT *>(P.coef);
    if constexpr (!HET)  // role-major in HBM -> role-interleaved 16-byte chunks in LDS
        for (int i = tid; i < G * PK::CP; i += T) s_coef[CoefLds<RT, G>::slot(i % PK::CP, i / PK::CP)] = gcoef[i];
    const int bnd_len = N * G * PK::BW + G * PK::DW;
    for (int i = tid; i < bnd_len; i += T) s_bnd[i] = P.bounds[i];
    __syncthreads();

    const int q = tid % G;
    const long B = P.batch, BG = G * B;
    const long slot = (long)blockIdx.x * (T / G) + tid / G;
    const bool active = slot < B;
    const long bb = active ? slot : 0;                      // dense index: scratch columns (and families, HET)
    const long b = (active && P.idx) ? P.idx[slot] : slot;  // instance this lane group works on
    const long L = G * bb + q;  // this lane's column in the per-instance coefficient matrix (HET)
    const long EX = (long)NX * N, EU = (long)NU * (N - 1);
    const float *lb = s_bnd + q * PK::BW;

    float cQD[RX], cRD[RU];
    float rho = P.rho;
    if constexpr (HET) {  // per instance: het_aux = [Qd (nx) | Rd (nu) | rho][batch]
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
    using CPtr = std::conditional_t<HET, CoefCol<RT>, CoefRole<RT, G>>;
    CPtr cbase;
    if constexpr (HET)
        cbase = CoefCol<RT>{gcoef, BG, (unsigned)(L * sizeof(RT))};
    else
        cbase = CoefRole<RT, G>{s_coef + q * CoefRole<RT, G>::VEC, 0};
    const CPtr cA = cbase + S::O_A, cB = cbase + S::O_B, cAT = cbase + S::O_AT, cBT = cbase + S::O_BT, cQI = cbase + S::O_QI,
               cF = cbase + PK::O_F, cAPF = cbase + PK::O_APF, cBPF = cbase + PK::O_BPF, cATT = cbase + PK::O_ATT;
    (void)cATT;
    using KPtr = std::conditional_t<ADP, CoefCol<RT>, CPtr>;   // the rows made of Kinf / Pinf
    KPtr cK, cKT, cPT;
    RT *const adp_cols = ADP ? reinterpret_cast<RT *>(P.adp_cols) : nullptr;
    double rho_d = (double)P.rho;                           // ADP: this instance's rho as the adaptive state holds it
    if constexpr (ADP) {
        const CoefCol<RT> col{adp_cols, BG, (unsigned)(L * sizeof(RT))};
        cK = col + PK::AO_K, cKT = col + PK::AO_KT, cPT = col + PK::AO_PT;
        // fill the column from the solver's adaptive state [1 + nu nx + nx nx][batch] (rho | Kinf | Pinf, column-major)
        const long AB = P.adapt_stride;
        const double *ad = P.adapt + b;
        if (active) {
            rho_d = ad[0];
            rho = (float)rho_d;
            RT *mine = adp_cols + L;
#pragma unroll
            for (int m = 0; m < RU; ++m)
                for (int j = 0; j < NXP; ++j) {
                    const int a = q * RU + m;
                    mine[(long)(PK::AO_K + m * NXP + j) * BG] = (a < NU && j < NX) ? (RT)ad[(long)(1 + a + j * NU) * AB] : (RT)0;
                }
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int r = q * RX + m;
                for (int a = 0; a < NUP; ++a)
                    mine[(long)(PK::AO_KT + m * NUP + a) * BG] = (r < NX && a < NU) ? (RT)ad[(long)(1 + a + r * NU) * AB] : (RT)0;
                for (int j = 0; j < NXP; ++j)
                    mine[(long)(PK::AO_PT + m * NXP + j) * BG] =
                        (r < NX && j < NX) ? (RT)ad[(long)(1 + NU * NX + j + r * NX) * AB] : (RT)0;   // Pinf^T[r][j] = Pinf[j][r]
            }
        }
    } else {
        cK = cbase + S::O_K, cKT = cbase + S::O_KT, cPT = cbase + S::O_PT;
    }

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
    const int mlx = EXT == 2 ? P.mlx : 0, mlu = EXT == 2 ? P.mlu : 0;
    if constexpr (EXT == 2) {
        const float *gAx = P.lin, *gbx = gAx + mlx * NX, *gn2x = gbx + mlx;
        const float *gAu = gn2x + mlx, *gbu = gAu + mlu * NU, *gn2u = gbu + mlu;
        for (int i = tid; i < LIN_MAX_ROWS * G * RX; i += T) {
            const int k = i / (G * RX), row = i % (G * RX);  // (role, local row) flattened = padded row index
            s_lax[i] = (k < mlx && row < NX) ? gAx[k * NX + row] : 0.f;
        }
        for (int i = tid; i < LIN_MAX_ROWS * G * RU; i += T) {
            const int k = i / (G * RU), row = i % (G * RU);
            s_lau[i] = (k < mlu && row < NU) ? gAu[k * NU + row] : 0.f;
        }
        if (tid < LIN_MAX_ROWS) {
            s_lb[tid] = tid < mlx ? gbx[tid] : 0.f;
            s_lb[LIN_MAX_ROWS + tid] = tid < mlx ? gn2x[tid] : 1.f;
            s_lb[2 * LIN_MAX_ROWS + tid] = tid < mlu ? gbu[tid] : 0.f;
            s_lb[3 * LIN_MAX_ROWS + tid] = tid < mlu ? gn2u[tid] : 1.f;
        }
        __syncthreads();
    }
    const bool lin_x = mlx > 0, lin_u = mlu > 0;
    const float *lax = s_lax + q * RX, *lau = s_lau + q * RU;

    // Scratch: [array][knot][instance][real row].  Addresses are (uniform 64-bit base in SGPRs) + (32-bit byte
    // offset of the lane); the knot index is made opaque per knot (knot_sgpr) so that no per-array 64-bit
    // pointers are carried through the sweeps in VGPRs.  Lanes / rows beyond nx, nu read a clamped (valid)
    // address, the consumer discards the value (mkx / mku), and they do not store.
    const long BNX = B * NX, BNU = B * NU;
    const long SXN = BNX * N, SUN = BNU * (N - 1);
    const bool xl = q < NXL, ul = q < NUL;
    unsigned okx = 0u, oku = 0u, lxo[RX], luo[RU];
#pragma unroll
    for (int m = 0; m < RX; ++m) {
        const int row = q * RX + m;
        if (row < NX) okx |= 1u << m;
        const int rc = XFULL ? (q < NXL ? q : NXL - 1) * RX : (row < NX ? row : NX - 1);
        lxo[m] = (unsigned)((bb * NX + rc) * 4);
    }
#pragma unroll
    for (int m = 0; m < RU; ++m) {
        const int row = q * RU + m;
        if (row < NU) oku |= 1u << m;
        const int rc = UFULL ? (q < NUL ? q : NUL - 1) * RU : (row < NU ? row : NU - 1);
        luo[m] = (unsigned)((bb * NU + rc) * 4);
    }
    float *const Sg = P.scratch, *const Sw = Sg + SXN, *const Sv = Sw + SXN;
    float *const Sy = Sv + SXN, *const Szw = Sy + SUN, *const Sz = Szw + SUN, *const Sd = Sz + SUN;
    float *const Sgc = Sd + SUN, *const Swc = Sgc + SXN, *const Svc = Swc + SXN;
    float *const Syc = Svc + SXN, *const Szwc = Syc + SUN, *const Szc = Szwc + SUN;
    float *const Sgl = Szc + SUN, *const Swl = Sgl + SXN, *const Svl = Swl + SXN;
    float *const Syl = Svl + SXN, *const Szwl = Syl + SUN, *const Szl = Szwl + SUN;
    float *const Ss = Sv, *const Ssu = Sz;  // OS: the fused arrays live where v and z would
#define SXP(arr, k, m) lane_elem(sgpr_ptr(arr + ((long)(k)*BNX + (XFULL ? (m) : 0))), lxo[XFULL ? 0 : (m)])
#define SUP(arr, k, m) lane_elem(sgpr_ptr(arr + ((long)(k)*BNU + (UFULL ? (m) : 0))), luo[UFULL ? 0 : (m)])
#define OKX(m) (ALLX || (XFULL ? xl : (((okx >> (m)) & 1u) != 0u)))
#define OKU(m) (ALLU || (UFULL ? ul : (((oku >> (m)) & 1u) != 0u)))
    auto ldx = [&](float *arr, int k, float (&dst)[RX]) {
        if constexpr (XFULL) {
            load_rows<RX>(SXP(arr, k, 0), dst);
        } else {
#pragma unroll
            for (int m = 0; m < RX; ++m) dst[m] = *SXP(arr, k, m);
        }
    };
    auto ldu = [&](float *arr, int k, float (&dst)[RU]) {
        if constexpr (UFULL) {
            load_rows<RU>(SUP(arr, k, 0), dst);
        } else {
#pragma unroll
            for (int m = 0; m < RU; ++m) dst[m] = *SUP(arr, k, m);
        }
    };
    // stores of one array's rows; the caller has already excluded lanes that own no row (XFULL / UFULL shapes)
    auto stx = [&](float *arr, int k, const float (&src)[RX]) {
        if constexpr (XFULL) {
            store_rows<RX>(SXP(arr, k, 0), src);
        } else {
#pragma unroll
            for (int m = 0; m < RX; ++m)
                if (OKX(m)) *SXP(arr, k, m) = src[m];
        }
    };
    auto stu = [&](float *arr, int k, const float (&src)[RU]) {
        if constexpr (UFULL) {
            store_rows<RU>(SUP(arr, k, 0), src);
        } else {
#pragma unroll
            for (int m = 0; m < RU; ++m)
                if (OKU(m)) *SUP(arr, k, m) = src[m];
        }
    };
    auto mkx = [&](float v, int m) __attribute__((always_inline)) { return OKX(m) ? v : 0.f; };
    auto mku = [&](float v, int m) __attribute__((always_inline)) { return OKU(m) ? v : 0.f; };
    const bool x_owner = ALLX || !XFULL || xl, u_owner = ALLU || !UFULL || ul;
    // One-shot solve without a finite state bound: vnew = x + g is never clamped, so the state dual stays at its
    // cold-start value, zero — it is neither streamed nor computed (bit-identical results).  Plain kernels only: with
    // the extensions compiled in, the extra run-time case costs config 4 (which has finite state bounds) 3 %.
    const bool nog = EXT == 0 && OS && !P.xb_active;

    RT x0[RX];
#pragma unroll
    for (int m = 0; m < RX; ++m) x0[m] = (active && q * RX + m < NX) ? (RT)P.x0[b * NX + q * RX + m] : (RT)0;
    const bool warm = active && !P.cold_start;
    if (active) {
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                if (row >= NX) continue;
                *SXP(Sg, k, m) = warm ? P.sg[b * EX + k * NX + row] : 0.f;
                *SXP(Sw, k, m) = 0.f;
                if constexpr (!OS) *SXP(Sv, k, m) = warm ? P.sv[b * EX + k * NX + row] : 0.f;
                if (soc_x) {
                    *SXP(Sgc, k, m) = warm ? P.sgc[b * EX + k * NX + row] : 0.f;
                    *SXP(Swc, k, m) = 0.f;
                    if constexpr (!OS) *SXP(Svc, k, m) = warm ? P.svc[b * EX + k * NX + row] : 0.f;
                }
                if (lin_x) {
                    *SXP(Sgl, k, m) = warm ? P.sgl[b * EX + k * NX + row] : 0.f;
                    *SXP(Swl, k, m) = 0.f;
                    if constexpr (!OS) *SXP(Svl, k, m) = warm ? P.svl[b * EX + k * NX + row] : 0.f;
                }
            }
        for (int k = 0; k < N - 1; ++k)
#pragma unroll
            for (int m = 0; m < RU; ++m) {
                const int row = q * RU + m;
                if (row >= NU) continue;
                *SUP(Sy, k, m) = warm ? P.sy[b * EU + k * NU + row] : 0.f;
                *SUP(Sd, k, m) = warm ? P.sd[b * EU + k * NU + row] : 0.f;
                *SUP(Szw, k, m) = 0.f;
                if constexpr (!OS) *SUP(Sz, k, m) = warm ? P.sz[b * EU + k * NU + row] : 0.f;
                if (soc_u) {
                    *SUP(Syc, k, m) = warm ? P.syc[b * EU + k * NU + row] : 0.f;
                    *SUP(Szwc, k, m) = 0.f;
                    if constexpr (!OS) *SUP(Szc, k, m) = warm ? P.szc[b * EU + k * NU + row] : 0.f;
                }
                if (lin_u) {
                    *SUP(Syl, k, m) = warm ? P.syl[b * EU + k * NU + row] : 0.f;
                    *SUP(Szwl, k, m) = 0.f;
                    if constexpr (!OS) *SUP(Szl, k, m) = warm ? P.szl[b * EU + k * NU + row] : 0.f;
                }
            }
    }
    auto ref_x = [&](int k, int m) __attribute__((always_inline)) -> float {
        const int row = q * RX + m;
        if (row >= NX || P.ref_mode == REF_ZERO) return 0.f;
        return P.ref_mode == REF_SHARED ? P.xref[k * NX + row] : P.xref[b * EX + k * NX + row];
    };
    auto ref_u = [&](int k, int m) __attribute__((always_inline)) -> float {
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
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;
    // where the previous iteration's vnew / znew are found (dual residual, admm.cpp:93-96)
    float *const Svold = OS ? Sw : Sv, *const Svcold = OS ? Swc : Svc, *const Szold = OS ? Szw : Sz,
                 *const Szcold = OS ? Szwc : Szc, *const Svlold = OS ? Swl : Svl, *const Szlold = OS ? Szwl : Szl;

    // what one knot of the forward / backward sweep reads from the scratch block, D knots ahead of its use
    struct FwdBuf {
        float g[RX], v[RX], gc[RX], vc[RX], gl[RX], vl[RX], d[RU], y[RU], z[RU], yc[RU], zc[RU], yl[RU], zl[RU];
    };
    struct BwdBuf {  // OS: the fused arrays in w / zw; else all of them
        float w[RX], g[RX], wc[RX], gc[RX], wl[RX], gl[RX], zw[RU], y[RU], zwc[RU], yc[RU], zwl[RU], yl[RU];
    };

    for (int i = 0; i < P.max_iter; ++i) {
        if (active && !conv) {
            const bool check = ct > 0 && (i + 1) % ct == 0;  // lanes still iterating have it == i: wave-uniform
            const bool need_res = check && (can_converge || i + 1 == last_check_it);
            // OS: vnew / znew themselves are only read back as "previous" values by a residual check and as the
            // solution, so they are stored when this iteration may be a lane's last or the next one checks
            const bool check_next = ct > 0 && (i + 2) % ct == 0;
            const bool keep_w = !OS || need_res || i + 1 == P.max_iter ||
                                (check_next && (can_converge || i + 2 == last_check_it));
            // ================= fused forward sweep (admm.cpp:25-69 + :93-96) =================
            RT x[RX];
#pragma unroll
            for (int m = 0; m < RX; ++m) x[m] = x0[m];
            float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
            // adaptive rho: the iterations that adapt (admm.cpp:147) gather the norms of rho_benchmark.cpp:44-213 on the
            // way — constraint rows [u_k; A x_k + B u_k (+ f) - x_{k+1}] against [znew_k; vnew_{k+1}], cost rows
            // P x + q + A'y with P = blkdiag(Q~, R~, .., Pinf), q = [Q~ x; R~ u] (zero reference), y = [y_k; g_{k+1}] —
            // each knot contributing the rows of knot k - 1 that needed its new dual g_k.
            const bool adapt_now = ADP && i > 0 && i % 5 == 0;
            const float rho_lin = rho;   // the linear cost of this iteration is formed before the adaptation (admm.cpp:139 vs :147)
            RT a_pri = 0, a_axm = 0, a_zm = 0, a_dres = 0, a_pxm = 0, a_atym = 0, a_qm = 0;
            RT a_xp[RX];
            float a_gp[RX], a_up[RU], a_yp[RU];              // knot k - 1: x, new g, u, new y
#pragma unroll
            for (int m = 0; m < RX; ++m) a_xp[m] = (RT)0, a_gp[m] = 0.f;
#pragma unroll
            for (int m = 0; m < RU; ++m) a_up[m] = a_yp[m] = 0.f;
            FwdBuf fb[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
#pragma unroll
                for (int m = 0; m < RX; ++m)
                    fb[j].g[m] = fb[j].v[m] = fb[j].gc[m] = fb[j].vc[m] = fb[j].gl[m] = fb[j].vl[m] = 0.f;
#pragma unroll
                for (int m = 0; m < RU; ++m)
                    fb[j].d[m] = fb[j].y[m] = fb[j].z[m] = fb[j].yc[m] = fb[j].zc[m] = fb[j].yl[m] = fb[j].zl[m] = 0.f;
            }
            auto fetch_x = [&](int k_, FwdBuf &f) __attribute__((always_inline)) {
                const int k = knot_sgpr(k_);
                if (!nog) ldx(Sg, k, f.g);
                if (need_res) ldx(Svold, k, f.v);
                if (soc_x) {
                    ldx(Sgc, k, f.gc);
                    if (need_res) ldx(Svcold, k, f.vc);
                }
                if (lin_x) {
                    ldx(Sgl, k, f.gl);
                    if (need_res) ldx(Svlold, k, f.vl);
                }
            };
            auto fetch_u = [&](int k_, FwdBuf &f) __attribute__((always_inline)) {
                const int k = knot_sgpr(k_);
                ldu(Sd, k, f.d);
                ldu(Sy, k, f.y);
                if (need_res) ldu(Szold, k, f.z);
                if (soc_u) {
                    ldu(Syc, k, f.yc);
                    if (need_res) ldu(Szcold, k, f.zc);
                }
                if (lin_u) {
                    ldu(Syl, k, f.yl);
                    if (need_res) ldu(Szlold, k, f.zl);
                }
            };
            auto fwd_knot = [&](int k_, FwdBuf &f) __attribute__((always_inline)) {
                asm volatile("" ::: "memory");  // keep coefficient / bound loads per knot (no hoisting into registers)
                const int k = knot_sgpr(k_);
                const bool pf = k + D < N;  // this knot's buffer is refilled for knot k + D once consumed
                const float *bk = lb + k * G * PK::BW;
                float xf[RX], vn[RX], gn[RX], wc[RX], gcn[RX], wl[RX], gln[RX], sx[RX];
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    const float g_c = nog ? 0.f : mkx(f.g[m], m), v_c = mkx(f.v[m], m);
                    xf[m] = (float)x[m];
                    vn[m] = fminf(bk[RX + m], fmaxf(bk[m], xf[m] + g_c));
                    gn[m] = (g_c + xf[m]) - vn[m];
                    pri_x = fmaxf(pri_x, fabsf(xf[m] - vn[m]));
                    dua_x = fmaxf(dua_x, fabsf(v_c - vn[m]));
                    sx[m] = vn[m] - gn[m];
                }
                if constexpr (EXT) {
                    if (soc_x) {
#pragma unroll
                        for (int m = 0; m < RX; ++m) wc[m] = xf[m] + mkx(f.gc[m], m);
                        for (int c = 0; c < ncx; ++c) {
                            const uint4 mk = cm[c * G];
                            project_soc_group<G, RX>(wc, mk.x, mk.y, P.cx[c]);
                        }
#pragma unroll
                        for (int m = 0; m < RX; ++m) {
                            gcn[m] = (mkx(f.gc[m], m) + xf[m]) - wc[m];
                            pri_x = fmaxf(pri_x, fabsf(xf[m] - wc[m]));
                            dua_x = fmaxf(dua_x, fabsf(mkx(f.vc[m], m) - wc[m]));
                            sx[m] += wc[m] - gcn[m];
                        }
                    }
                }
                if constexpr (EXT == 2) {
                    if (lin_x) {
#pragma unroll
                        for (int m = 0; m < RX; ++m) wl[m] = xf[m] + mkx(f.gl[m], m);
                        project_halfspaces_group<G, RX>(wl, lax, G * RX, mlx, s_lb, s_lb + LIN_MAX_ROWS);
#pragma unroll
                        for (int m = 0; m < RX; ++m) {
                            gln[m] = (mkx(f.gl[m], m) + xf[m]) - wl[m];
                            pri_x = fmaxf(pri_x, fabsf(xf[m] - wl[m]));
                            dua_x = fmaxf(dua_x, fabsf(mkx(f.vl[m], m) - wl[m]));
                            sx[m] += wl[m] - gln[m];
                        }
                    }
                }
                if (x_owner) {
                    if (!nog) stx(Sg, k, gn);
                    if (keep_w) stx(Sw, k, vn);
                    if constexpr (OS) stx(Ss, k, sx);
                    if constexpr (EXT)
                        if (soc_x) {
                            stx(Sgc, k, gcn);
                            if (keep_w) stx(Swc, k, wc);
                        }
                    if constexpr (EXT == 2)
                        if (lin_x) {
                            stx(Sgl, k, gln);
                            if (keep_w) stx(Swl, k, wl);
                        }
                }
                if constexpr (ADP) {
                    if (adapt_now) {
                        RT gk[RX], xfl[RX];
#pragma unroll
                        for (int m = 0; m < RX; ++m) gk[m] = (RT)gn[m], xfl[m] = (RT)xf[m];
                        if (k >= 1) {
                            RT atx[RX], atu[RU];
#pragma unroll
                            for (int m = 0; m < RX; ++m) atx[m] = (RT)0;
#pragma unroll
                            for (int m = 0; m < RU; ++m) atu[m] = (RT)0;
                            quad_matvec<G, RX, NXL, RX, NXP>(atx, cATT, gk);     // A' g_k
                            quad_matvec<G, RU, NXL, RX, NXP>(atu, cBT, gk);      // B' g_k
#pragma unroll
                            for (int m = 0; m < RX; ++m) {                       // state rows of knot k - 1
                                if (k >= 2) atx[m] -= (RT)a_gp[m];
                                const RT qv = (RT)cQD[m] * a_xp[m];
                                upmax_abs(a_dres, qv + qv + atx[m]);
                                upmax_abs(a_pxm, qv);
                                upmax_abs(a_qm, qv);
                                upmax_abs(a_atym, atx[m]);
                                // the rollout's own x_k makes A x + B u (+ f) - x_k vanish: the row is 0 against vnew_k
                                upmax_abs(a_pri, (RT)vn[m]);
                                upmax_abs(a_zm, (RT)vn[m]);
                            }
#pragma unroll
                            for (int m = 0; m < RU; ++m) {                       // input rows of knot k - 1
                                const RT px = (RT)cRD[m] * (RT)a_up[m], aty = (RT)a_yp[m] + atu[m];
                                upmax_abs(a_dres, px + px + aty);
                                upmax_abs(a_pxm, px);
                                upmax_abs(a_qm, px);
                                upmax_abs(a_atym, aty);
                            }
                        }
                        if (k == N - 1) {                                        // the terminal knot's own rows: Pinf x + Q~ x - g
                            RT px[RX];
#pragma unroll
                            for (int m = 0; m < RX; ++m) px[m] = (RT)0;
                            quad_matvec<G, RX, NXL, RX, NXP>(px, cPT, xfl);
#pragma unroll
                            for (int m = 0; m < RX; ++m) {
                                const RT qv = (RT)cQD[m] * xfl[m], aty = (k >= 1) ? -gk[m] : (RT)0;
                                upmax_abs(a_dres, px[m] + qv + aty);
                                upmax_abs(a_pxm, px[m]);
                                upmax_abs(a_qm, qv);
                                upmax_abs(a_atym, aty);
                            }
                        }
#pragma unroll
                        for (int m = 0; m < RX; ++m) a_xp[m] = xfl[m], a_gp[m] = gn[m];
                    }
                }
                if (pf) fetch_x(k + D, f);
                if (k < N - 1) {
                    RT u[RU], xn[RX];
                    float uf[RU], zn[RU], yn[RU], zc2[RU], ycn[RU], zl2[RU], yln[RU], su[RU];
#pragma unroll
                    for (int m = 0; m < RU; ++m) u[m] = -(RT)mku(f.d[m], m);
#pragma unroll
                    for (int m = 0; m < RX; ++m) xn[m] = EXT ? (RT)cF[m] : (RT)0;
                    asm volatile("" ::: "memory");  // (and between products: each one's coefficient loads stay next to their use)
                    quad_matvec<G, RU, NXL, RX, NXP, true>(u, cK, x);    // u = -d - Kinf x
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RX, NXL, RX, NXP>(xn, cA, x);   // A x (+ fdyn)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        const float y_c = mku(f.y[m], m);
                        uf[m] = (float)u[m];
                        zn[m] = fminf(bk[2 * RX + RU + m], fmaxf(bk[2 * RX + m], uf[m] + y_c));
                        yn[m] = (y_c + uf[m]) - zn[m];
                        pri_u = fmaxf(pri_u, fabsf(uf[m] - zn[m]));
                        dua_u = fmaxf(dua_u, fabsf(mku(f.z[m], m) - zn[m]));
                        su[m] = zn[m] - yn[m];
                    }
                    if constexpr (EXT) {
                        if (soc_u) {
#pragma unroll
                            for (int m = 0; m < RU; ++m) zc2[m] = uf[m] + mku(f.yc[m], m);
                            for (int c = 0; c < ncu; ++c) {
                                const uint4 mk = cm[c * G];
                                project_soc_group<G, RU>(zc2, mk.z, mk.w, P.cu[c]);
                            }
#pragma unroll
                            for (int m = 0; m < RU; ++m) {
                                ycn[m] = (mku(f.yc[m], m) + uf[m]) - zc2[m];
                                pri_u = fmaxf(pri_u, fabsf(uf[m] - zc2[m]));
                                dua_u = fmaxf(dua_u, fabsf(mku(f.zc[m], m) - zc2[m]));
                                su[m] += zc2[m] - ycn[m];
                            }
                        }
                    }
                    if constexpr (EXT == 2) {
                        if (lin_u) {
#pragma unroll
                            for (int m = 0; m < RU; ++m) zl2[m] = uf[m] + mku(f.yl[m], m);
                            project_halfspaces_group<G, RU>(zl2, lau, G * RU, mlu, s_lb + 2 * LIN_MAX_ROWS,
                                                            s_lb + 3 * LIN_MAX_ROWS);
#pragma unroll
                            for (int m = 0; m < RU; ++m) {
                                yln[m] = (mku(f.yl[m], m) + uf[m]) - zl2[m];
                                pri_u = fmaxf(pri_u, fabsf(uf[m] - zl2[m]));
                                dua_u = fmaxf(dua_u, fabsf(mku(f.zl[m], m) - zl2[m]));
                                su[m] += zl2[m] - yln[m];
                            }
                        }
                    }
                    if (u_owner) {
                        stu(Sy, k, yn);
                        if (keep_w) stu(Szw, k, zn);
                        if constexpr (OS) stu(Ssu, k, su);
                        if constexpr (EXT)
                            if (soc_u) {
                                stu(Syc, k, ycn);
                                if (keep_w) stu(Szwc, k, zc2);
                            }
                        if constexpr (EXT == 2)
                            if (lin_u) {
                                stu(Syl, k, yln);
                                if (keep_w) stu(Szwl, k, zl2);
                            }
                    }
                    if constexpr (ADP) {
                        if (adapt_now) {
#pragma unroll
                            for (int m = 0; m < RU; ++m) {
                                upmax_abs(a_pri, (RT)uf[m] - (RT)zn[m]);
                                upmax_abs(a_axm, (RT)uf[m]);
                                upmax_abs(a_zm, (RT)zn[m]);
                                a_up[m] = uf[m], a_yp[m] = yn[m];
                            }
                        }
                    }
                    if (pf && k + D < N - 1) fetch_u(k + D, f);
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RX, NUL, RU, NUP>(xn, cB, u);   // + B u
#pragma unroll
                    for (int m = 0; m < RX; ++m) x[m] = xn[m];
                }
            };
            sfor<0, D>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if (j < N) {
                    fetch_x(j, fb[j]);
                    if (j < N - 1) fetch_u(j, fb[j]);
                }
            });
            for (int k0 = 0; k0 < N; k0 += D)
                sfor<0, D>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    if (k0 + j < N) fwd_knot(k0 + j, fb[j]);
                });
            it += 1;
            RT accP[RX];                                     // Pinf' xref_{N-1}, with the Pinf the linear cost was formed with
#pragma unroll
            for (int m = 0; m < RX; ++m) accP[m] = (RT)0;
            if constexpr (ADP) {
                if (P.ref_mode != REF_ZERO) {
                    RT xrl[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) xrl[m] = (RT)ref_x(N - 1, m);
                    quad_matvec<G, RX, NXL, RX, NXP>(accP, cPT, xrl);
                }
                if (adapt_now) {
                    // predict_rho (rho_benchmark.cpp:173-195), then the first-order update of Kinf, Pinf (admm.cpp:160-172)
                    const RT pri = group_max_t<G>(a_pri), axm = group_max_t<G>(a_axm), zm = group_max_t<G>(a_zm),
                             dres = group_max_t<G>(a_dres), pxm = group_max_t<G>(a_pxm), atym = group_max_t<G>(a_atym),
                             qm = group_max_t<G>(a_qm);
                    const RT eps = (RT)1e-10, prin = axm > zm ? axm : zm;
                    RT duan = pxm > atym ? pxm : atym;
                    duan = qm > duan ? qm : duan;
                    const RT ratio = (pri / (prin + eps)) / (dres / (duan + eps) + eps);
                    RT nrho = (RT)rho_d * (RT)sqrt((double)ratio);
                    if (P.rho_clip) nrho = nrho < (RT)P.rho_min ? (RT)P.rho_min : (nrho > (RT)P.rho_max ? (RT)P.rho_max : nrho);
                    const double delta = (double)nrho - rho_d;
                    const long AB = P.adapt_stride;
                    double *ad = P.adapt + b;
                    RT *mine = adp_cols + L;
                    const double *sK = P.sens, *sP = P.sens + NU * NX;
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        const int a = q * RU + m;
                        if (a < NU)
                            for (int j = 0; j < NX; ++j) {       // this lane owns row a of Kinf: the solver's state and its own copy
                                const double v = ad[(long)(1 + a + j * NU) * AB] + delta * sK[a + j * NU];
                                ad[(long)(1 + a + j * NU) * AB] = v;
                                mine[(long)(PK::AO_K + m * NXP + j) * BG] = (RT)v;
                            }
                    }
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        const int r = q * RX + m;
                        if (r < NX) {
                            for (int a = 0; a < NU; ++a) {       // Kinf^T rows: another lane owns the state's copy
                                RT *c = mine + (long)(PK::AO_KT + m * NUP + a) * BG;
                                *c = (RT)((double)*c + delta * sK[a + r * NU]);
                            }
                            for (int j = 0; j < NX; ++j) {       // column r of Pinf
                                const double v = ad[(long)(1 + NU * NX + j + r * NX) * AB] + delta * sP[j + r * NX];
                                ad[(long)(1 + NU * NX + j + r * NX) * AB] = v;
                                mine[(long)(PK::AO_PT + m * NXP + j) * BG] = (RT)v;
                            }
                        }
                    }
                    if (q == 0) ad[0] = (double)nrho;
                    rho_d = (double)nrho;
                    rho = (float)nrho;
                }
            }
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
                // per knot it needs vnew - g (+ the cone set's) and znew - y (+ ...): the fused arrays in OS mode,
                // else formed from w, g, (wc, gc) while copying w -> v, zw -> z (admm.cpp:196-197)
                auto fetchb_x = [&](int k_, BwdBuf &f) __attribute__((always_inline)) {
                    const int k = knot_sgpr(k_);
                    if constexpr (OS) {
                        ldx(Ss, k, f.w);
                    } else {
                        ldx(Sw, k, f.w);
                        ldx(Sg, k, f.g);
                        if (soc_x) {
                            ldx(Swc, k, f.wc);
                            ldx(Sgc, k, f.gc);
                        }
                        if (lin_x) {
                            ldx(Swl, k, f.wl);
                            ldx(Sgl, k, f.gl);
                        }
                    }
                };
                auto fetchb_u = [&](int k_, BwdBuf &f) __attribute__((always_inline)) {
                    const int k = knot_sgpr(k_);
                    if constexpr (OS) {
                        ldu(Ssu, k, f.zw);
                    } else {
                        ldu(Szw, k, f.zw);
                        ldu(Sy, k, f.y);
                        if (soc_u) {
                            ldu(Szwc, k, f.zwc);
                            ldu(Syc, k, f.yc);
                        }
                        if (lin_u) {
                            ldu(Szwl, k, f.zwl);
                            ldu(Syl, k, f.yl);
                        }
                    }
                };
                // consume the state-shaped part of a buffer: returns rho-less (vnew - g [+ cone set]) per row
                auto take_x = [&](int k, BwdBuf &f, float (&sx)[RX]) {
                    if constexpr (OS) {
#pragma unroll
                        for (int m = 0; m < RX; ++m) sx[m] = mkx(f.w[m], m);
                    } else {
#pragma unroll
                        for (int m = 0; m < RX; ++m) {
                            sx[m] = mkx(f.w[m] - f.g[m], m);
                            if (soc_x) sx[m] += mkx(f.wc[m] - f.gc[m], m);
                            if (lin_x) sx[m] += mkx(f.wl[m] - f.gl[m], m);
                        }
                        if (x_owner) {
                            stx(Sv, k, f.w);
                            if (soc_x) stx(Svc, k, f.wc);
                            if (lin_x) stx(Svl, k, f.wl);
                        }
                    }
                };
                auto take_u = [&](int k, BwdBuf &f, float (&su)[RU]) {
                    if constexpr (OS) {
#pragma unroll
                        for (int m = 0; m < RU; ++m) su[m] = mku(f.zw[m], m);
                    } else {
#pragma unroll
                        for (int m = 0; m < RU; ++m) {
                            su[m] = mku(f.zw[m] - f.y[m], m);
                            if (soc_u) su[m] += mku(f.zwc[m] - f.yc[m], m);
                            if (lin_u) su[m] += mku(f.zwl[m] - f.yl[m], m);
                        }
                        if (u_owner) {
                            stu(Sz, k, f.zw);
                            if (soc_u) stu(Szc, k, f.zwc);
                            if (lin_u) stu(Szl, k, f.zwl);
                        }
                    }
                };
                BwdBuf bbuf[D], bterm;
                fetchb_x(N - 1, bterm);
                sfor<0, D>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    if (j < N - 1) {
                        fetchb_x(N - 2 - j, bbuf[j]);
                        fetchb_u(N - 2 - j, bbuf[j]);
                    }
                });
                RT p[RX];
                {
                    RT acc[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) acc[m] = accP[m];
                    if (!ADP && P.ref_mode != REF_ZERO) {
                        RT xrl[RX];
#pragma unroll
                        for (int m = 0; m < RX; ++m) xrl[m] = (RT)ref_x(N - 1, m);
                        quad_matvec<G, RX, NXL, RX, NXP>(acc, cPT, xrl);
                    }
                    float sx[RX];
                    take_x(N - 1, bterm, sx);
#pragma unroll
                    for (int m = 0; m < RX; ++m) p[m] = -acc[m] - (RT)(rho_lin * sx[m]);
                }
                auto bwd_knot = [&](int t_, BwdBuf &f) __attribute__((always_inline)) {  // t counts knots from N-2 downwards
                    asm volatile("" ::: "memory");
                    const int t = knot_sgpr(t_);
                    const int k = N - 2 - t;
                    RT r[RU], qk[RX];
                    float sx[RX], su[RU];
                    take_x(k, f, sx);
                    take_u(k, f, su);
#pragma unroll
                    for (int m = 0; m < RU; ++m) r[m] = (RT)(-(ref_u(k, m) * cRD[m]) - rho_lin * su[m]);
#pragma unroll
                    for (int m = 0; m < RX; ++m) qk[m] = (RT)(-(ref_x(k, m) * cQD[m]) - rho_lin * sx[m]);
                    if (t + D < N - 1) {
                        fetchb_x(k - D, f);
                        fetchb_u(k - D, f);
                    }
                    RT tt[RU], dn[RU], ap[RX];
                    float dnf[RU];
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        tt[m] = r[m] + (EXT ? (RT)cBPF[m] : (RT)0);
                        dn[m] = (RT)0;
                    }
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RU, NXL, RX, NXP>(tt, cBT, p);   // B^T p + r (+ BPf)
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RU, NUL, RU, NUP>(dn, cQI, tt);  // d = Quu_inv (...)
#pragma unroll
                    for (int m = 0; m < RU; ++m) dnf[m] = (float)dn[m];
                    if (u_owner) stu(Sd, k, dnf);
#pragma unroll
                    for (int m = 0; m < RX; ++m) ap[m] = qk[m] + (EXT ? (RT)cAPF[m] : (RT)0);
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RX, NUL, RU, NUP, true>(ap, cKT, r);   // q (+ APf) - Kinf^T r: does not wait for p
                    asm volatile("" ::: "memory");
                    quad_matvec<G, RX, NXL, RX, NXP>(ap, cAT, p);         // + AmBKt p
#pragma unroll
                    for (int m = 0; m < RX; ++m) p[m] = ap[m];
                };
                for (int t0 = 0; t0 < N - 1; t0 += D)
                    sfor<0, D>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        if (t0 + j < N - 1) bwd_knot(t0 + j, bbuf[j]);
                    });
            }
        }
        if (!__builtin_amdgcn_ballot_w64(active && !conv)) break;
    }

    if (active) {
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int m = 0; m < RX; ++m)
                if (q * RX + m < NX) P.xout[b * EX + k * NX + q * RX + m] = *SXP(Sw, k, m);
        for (int k = 0; k < N - 1; ++k)
#pragma unroll
            for (int m = 0; m < RU; ++m)
                if (q * RU + m < NU) P.uout[b * EU + k * NU + q * RU + m] = *SUP(Szw, k, m);
        if (q == 0) {
            P.iter[b] = P.iter_offset + it;
            P.solved[b] = conv;
            P.res[b * 4 + 0] = res0;
            P.res[b * 4 + 1] = res1;
            P.res[b * 4 + 2] = res2;
            P.res[b * 4 + 3] = res3;
        }
        if constexpr (!OS) {
            if (P.save_state) {
                for (int k = 0; k < N; ++k)
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        const int row = q * RX + m;
                        if (row < NX) {
                            P.sg[b * EX + k * NX + row] = *SXP(Sg, k, m);
                            P.sv[b * EX + k * NX + row] = *SXP(Sv, k, m);
                            if (soc_x) {
                                P.sgc[b * EX + k * NX + row] = *SXP(Sgc, k, m);
                                P.svc[b * EX + k * NX + row] = *SXP(Svc, k, m);
                            }
                            if (lin_x) {
                                P.sgl[b * EX + k * NX + row] = *SXP(Sgl, k, m);
                                P.svl[b * EX + k * NX + row] = *SXP(Svl, k, m);
                            }
                        }
                    }
                for (int k = 0; k < N - 1; ++k)
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        const int row = q * RU + m;
                        if (row < NU) {
                            P.sy[b * EU + k * NU + row] = *SUP(Sy, k, m);
                            P.sz[b * EU + k * NU + row] = *SUP(Sz, k, m);
                            P.sd[b * EU + k * NU + row] = *SUP(Sd, k, m);
                            if (soc_u) {
                                P.syc[b * EU + k * NU + row] = *SUP(Syc, k, m);
                                P.szc[b * EU + k * NU + row] = *SUP(Szc, k, m);
                            }
                            if (lin_u) {
                                P.syl[b * EU + k * NU + row] = *SUP(Syl, k, m);
                                P.szl[b * EU + k * NU + row] = *SUP(Szl, k, m);
                            }
                        }
                    }
            }
        }
    }
#undef SXP
#undef SUP
#undef OKX
#undef OKU
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
        fold_status(P, m0, m1, m2, m3, __popcll(unsolved), tid);
    }
}

}  // namespace tmpc
