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
//     broadcast — no LDS traffic, no ds_bpermute, no MFMA (4x4 .. 12x12 is not a
//     contraction worth a matrix core).
//   * precision: the two serial recurrences (rollout x_{k+1} = A x_k + B u_k and the
//     Riccati gradient p_k, d_k) run in RT = double on the fp64 VALU with fp64
//     coefficients; everything the ADMM iteration stores (g, v, vnew, y, z, znew, d) and
//     all elementwise steps are fp32.  All-fp32 (RT = float) is selectable; it is faster
//     but its worst instance misses the 1e-5 parity target (DESIGN.md "Precision").
//   * every trajectory the ADMM iteration carries lives on chip for the whole solve:
//     in VGPRs (horizon loops fully unrolled, statically indexed), with the two
//     "previous slack" arrays optionally parked in LDS for long horizons.  x, u, q, r, p
//     of the reference are never materialised: they are produced and consumed knot by
//     knot inside the two fused sweeps.
//   * family constants: coefficient rows sit in VGPRs (small shapes) or LDS (nx = 12);
//     per-knot bounds and shared references are staged once per workgroup in LDS.
//   * HBM traffic is the compulsory I/O only: x0 (+refs, +warm state) in, x/u/status
//     (+warm state) out.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "admm_params.h"

#ifndef TMPC_COEF_LDS_THRESHOLD_G1
#define TMPC_COEF_LDS_THRESHOLD_G1 100000
#endif

// LDS state arrays: 0 = [element][thread] (one bank per lane and element), 1 = [knot][thread][rows of the knot]
// (a lane's rows contiguous: one ds_read/write_b64/b128 per knot instead of one b32 per row).  Measured on MI355X:
// cartpole with its slack arrays forced into LDS 0.460 -> 0.447 ms (all in registers: 0.369), quadrotor
// 11.67 -> 12.13 ms, rocket N=50 unchanged; 0 ships.
#ifndef TMPC_LDS_ROWS
#define TMPC_LDS_ROWS 0
#endif
#ifndef TMPC_FENCE_LDS_MATVEC
#define TMPC_FENCE_LDS_MATVEC 0
#endif

namespace tmpc {

// G = lanes per problem instance (a "group"): 4 = one DPP quad, 2 = half a quad, 1 = one lane.
// BUD64 / BUD32: register budget per lane (fp64 / fp32 recurrences) the placement aims at when only one
// wavefront per SIMD fits; tuned per shape on MI355X ("Where state lives").
// LOOPV: which cheaper loop variants are built and used for one-shot solves of this shape (bit 0: vnew/znew
// in place, bit 1: additionally no per-lane guard when nobody can converge, bit 2: folded signs in the fp64
// recurrences, see FOLD); measured per shape.
template <int NX_, int NU_, int N_, int G_ = 4, int BUD64_ = 380, int BUD32_ = 380, int LOOPV_ = 0>
struct QuadShape {
    static constexpr int LOOPV = LOOPV_;
    static_assert(G_ == 1 || G_ == 2 || G_ == 4, "group size must be 1, 2 or 4 lanes");
    static constexpr int NX = NX_, NU = NU_, N = N_, G = G_;
    static constexpr int RX = (NX + G - 1) / G, RU = (NU + G - 1) / G;
    static constexpr int NXP = G * RX, NUP = G * RU;
    static constexpr int NXL = (NX + RX - 1) / RX;  // lanes of a group owning real x rows
    static constexpr int NUL = (NU + RU - 1) / RU;  // lanes of a group owning real u rows
    // nu == 1: every lane of the group carries the single input row (u, y, z, d replicated), so the
    // input-side products need no cross-lane traffic at all; the packs hold row 0 for every role.
    static constexpr bool UREP = (NU == 1);
    // Coefficient pack per lane role q (elements of RT); rows beyond nx/nu and columns
    // beyond nx/nu are zero.  Filled by the host (kernels.hip: build_quad_coef).
    // Three blocks, each padded to 8 elements so a block can be fetched with 8-element scalar
    // loads: what the forward sweep reads | what the backward sweep reads | the terminal-cost rows.
    static constexpr int pad8(int n) { return (n + 7) / 8 * 8; }
    static constexpr int O_A = 0;                          // A       rows [RX][NXP]
    static constexpr int O_K = O_A + RX * NXP;             // Kinf    rows [RU][NXP]
    static constexpr int O_B = O_K + RU * NXP;             // B       rows [RX][NUP]
    static constexpr int FWD_LEN = pad8(O_B + RX * NUP);
    static constexpr int O_AT = FWD_LEN;                   // AmBKt   rows [RX][NXP]
    static constexpr int O_BT = O_AT + RX * NXP;           // B^T     rows [RU][NXP]
    static constexpr int O_KT = O_BT + RU * NXP;           // Kinf^T  rows [RX][NUP]
    static constexpr int O_QI = O_KT + RX * NUP;           // Quu_inv rows [RU][NUP]
    static constexpr int BWD_LEN = pad8(O_QI + RU * NUP - O_AT);
    static constexpr int O_PT = O_AT + BWD_LEN;            // Pinf^T  rows [RX][NXP]
    static constexpr int PT_LEN = pad8(RX * NXP);
    static constexpr int CP = O_PT + PT_LEN;
    static constexpr int CP_LIVE = 3 * RX * NXP + 2 * RU * NXP + 2 * RX * NUP + RU * NUP;
    // fp32 side pack per role: diag(Q)+rho [RX], diag(R)+rho [RU]
    static constexpr int DW = RX + RU;
    // Bounds pack: [N][G roles][xmin[RX] xmax[RX] umin[RU] umax[RU]]
    static constexpr int BW = 2 * RX + 2 * RU;
    static constexpr int BOUNDS_LEN = N * G * BW + G * DW;  // + the diag pack at the end
    // Shared-reference pack: [N][G roles][xref[RX] uref[RU]]
    static constexpr int RW = RX + RU;
    static constexpr int REFS_LEN = N * G * RW;
    static constexpr int THREADS = 256;
    static constexpr int INST_PER_BLOCK = THREADS / G;
    // adaptive rho (ADP kernels): built where a lane's coefficient rows live in its own registers — 4 lanes per instance
    // and a pack small enough for VGPRs (the cartpole shapes): the adapted Kinf / Pinf rows are then just those registers
    // ... or, with ONE lane per instance (wave-uniform coefficients), as a correction: the instance's Kinf enters as
    // dK = Kinf_b - Kinf_family = (rho_b - rho_family) dKinf/drho (nu nx values per lane) next to the family's products, its
    // Pinf — only needed at the terminal knot of adapting iterations — as Pinf_family + (rho_b - rho_family) dPinf/drho.
    // Zero or shared references (the terminal reference term is then (Pinf_family' + (rho_b - rho_family) dPinf') xref: two
    // wave-uniform vectors, formed once); the norm rows' A'g, B'g read the forward block's A and B transposed — with
    // wave-uniform coefficients any element is as near as any other.
    static constexpr bool ADP_OK = (G == 4 && CP_LIVE * 2 <= 72) || G == 1;
    // ---- storage policy ----
    // ---- storage policy -------------------------------------------------------------------
    // Coefficient rows: VGPRs when small, LDS when their (live) register footprint would exceed
    // ~1/4 of the VGPR file; with one lane per instance (G = 1) they are wave-uniform values.
    template <class RT, int REFS>
    static constexpr int coef_regs() {  // Pinf^T is only read when a reference trajectory exists
        return (CP_LIVE - (REFS == REF_ZERO ? RX * NXP : 0)) * (int)(sizeof(RT) / 4);
    }
    template <class RT, int REFS>
    static constexpr bool coef_in_lds() {
        return coef_regs<RT, REFS>() > TMPC_COEF_LDS_THRESHOLD_G1 || (G > 1 && coef_regs<RT, REFS>() > 72);
    }
    // The seven per-instance trajectories (floats per lane) and where each lives.  Arrays are moved
    // to LDS ([element][thread], conflict-free) in order of how rarely an iteration touches them,
    // first to make room for two wavefronts per SIMD (<= 250 VGPRs, <= ~78 KB LDS per workgroup),
    // else to fit one wavefront per SIMD (VGPRs + AGPRs, <= ~150 KB LDS).
    static constexpr int STATE_REGS = 3 * RX * N + 4 * RU * (N - 1);  // floats of state per lane
    enum { A_V = 0, A_Z, A_ZW, A_D, A_Y, A_W, A_G, A_COUNT };
    struct Placement {
        bool lds[A_COUNT];
        int off[A_COUNT];  // float offset of the array inside the LDS state block, per thread-stride
        int lds_floats;    // floats per lane in LDS
    };
    static constexpr int arr_size(int a, bool oneshot, bool nog = false) {
        if (oneshot && (a == A_W || a == A_ZW)) return 0;  // vnew / znew overwrite v / z in place
        if (nog && a == A_G) return 0;                     // the state dual is identically zero
        return (a == A_V || a == A_W || a == A_G) ? RX * N : RU * (N - 1);
    }
    template <class RT, int REFS, bool OS, bool NOG = false, int EXTRA = 0>   // EXTRA: registers beyond the model's working set
    static constexpr Placement place() {
        Placement p{};
        const int fixed = ((coef_in_lds<RT, REFS>() || G == 1) ? 0 : coef_regs<RT, REFS>()) +  // G = 1: SGPRs
                         
                          (G == 1 ? 150 : (sizeof(RT) == 8 ? 60 : 45)) +  // working registers of a knot
                          (REFS == REF_PER_INSTANCE ? RX * N + RU * (N - 1) : 0) + EXTRA;
        const int total = (OS ? 2 : 3) * RX * N + (OS ? 3 : 4) * RU * (N - 1) - (NOG ? RX * N : 0);
        // (register budget, LDS floats per lane) for 2 waves/SIMD, then 1 wave/SIMD
        const int budget[2] = {250 - fixed, (sizeof(RT) == 8 ? BUD64_ : BUD32_) - fixed};
        const int cap[2] = {78, 150};
        // one lane per instance is picked for batches of about one wavefront per SIMD: only the second target applies
        for (int pass = (G == 1 ? 1 : 0); pass < 2; ++pass) {
            int regs = total, lds = 0;
            bool sel[A_COUNT] = {};
            for (int a = 0; a < A_COUNT && regs > budget[pass]; ++a) {
                if (arr_size(a, OS, NOG) == 0 || lds + arr_size(a, OS, NOG) > cap[pass]) continue;
                sel[a] = true;
                regs -= arr_size(a, OS, NOG);
                lds += arr_size(a, OS, NOG);
            }
            if (regs <= budget[pass] || pass == 1) {
                int o = 0;
                for (int a = 0; a < A_COUNT; ++a) {
                    p.lds[a] = sel[a];
                    p.off[a] = o;
                    if (sel[a]) o += arr_size(a, OS, NOG);
                }
                p.lds_floats = lds;
                return p;
            }
        }
        return p;
    }
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
__device__ __forceinline__ int dpp_quad_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __builtin_bit_cast(float, dpp_quad_i<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
    const int lo = dpp_quad_i<CTRL>(__double2loint(v));
    const int hi = dpp_quad_i<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// value held by lane S of the caller's group of G lanes
template <int G, int S, class T>
__device__ __forceinline__ T gbcast(T v) {
    if constexpr (G == 4)
        return dpp_quad<S * 0x55>(v);  // quad_perm:[S,S,S,S]
    else if constexpr (G == 2)
        return dpp_quad<S | (S << 2) | ((2 + S) << 4) | ((2 + S) << 6)>(v);  // quad_perm:[S,S,2+S,2+S]
    else
        return v;
}
template <int G>
__device__ __forceinline__ float group_max(float m) {
    if constexpr (G >= 2) m = fmaxf(m, dpp_quad<0xB1>(m));  // quad_perm:[1,0,3,2]
    if constexpr (G == 4) m = fmaxf(m, dpp_quad<0x4E>(m));  // quad_perm:[2,3,0,1]
    return m;
}
// LDS image of the coefficient pack: 16-byte chunks of one role, the 4 roles of a chunk adjacent
// (64 contiguous bytes), so the 4 distinct addresses a wavefront reads per instruction fall in
// different banks.  (Role-major, as the pack is in HBM, puts all 4 roles on the same banks whenever
// the per-role size is a multiple of 256 B — a 4-way conflict on every coefficient read.)
template <class RT, int G>
struct CoefLds {
    static constexpr int VEC = 16 / (int)sizeof(RT);
    const RT *base;  // s_coef + q * VEC
    __device__ __forceinline__ static int slot(int i, int q) { return ((i / VEC) * G + q) * VEC + i % VEC; }
    __device__ __forceinline__ RT operator[](int i) const { return base[(i / VEC) * G * VEC + i % VEC]; }
    __device__ __forceinline__ CoefLds operator+(int off) const {
        // offsets used are multiples of VEC (every pack section is a multiple of G rows), so chunking commutes
        return CoefLds{base + (off / VEC) * G * VEC};
    }
};
// G = 1: a lane is an instance, so the coefficient rows are wave-uniform.  Each sweep fetches the
// block it needs with scalar loads into SGPRs (8 elements per s_load) and its FMAs read them as
// scalar operands — no VGPRs, no LDS, no per-use moves.  (Left to the compiler, the rows are kept
// for the whole kernel, overflow the ~100 SGPRs and come back through v_readlane on every use.)
template <class RT, int NLOADS>
struct SBlock {
    using V8 = RT __attribute__((ext_vector_type(8)));
    V8 c[NLOADS];
    __device__ __forceinline__ explicit SBlock(const RT *p) {
        static_assert(NLOADS >= 1 && NLOADS <= 6, "coefficient block too large for SGPRs");
        // one asm statement: every load, then the wait, so no output is consumed early
        if constexpr (sizeof(RT) == 8) {
            if constexpr (NLOADS == 2)
                asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(c[0]), "=&s"(c[1]) : "s"(p) : "memory");
            else if constexpr (NLOADS == 3)
                asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\t"
                             "s_load_dwordx16 %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]) : "s"(p) : "memory");
            else if constexpr (NLOADS == 4)
                asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\t"
                             "s_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3]) : "s"(p) : "memory");
            else
                for (int i = 0; i < NLOADS; ++i)
                    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(c[i]) : "s"(p + 8 * i) : "memory");
        } else {
            for (int i = 0; i < NLOADS; ++i)
                asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(c[i]) : "s"(p + 8 * i) : "memory");
        }
    }
    struct View {
        const SBlock &b;
        int off;
        __device__ __forceinline__ RT operator[](int i) const { return b.c[(off + i) / 8][(off + i) % 8]; }
    };
    __device__ __forceinline__ View at(int off) const { return View{*this, off}; }
};
// One-lane-per-instance kernels: a wavefront's solution through LDS to HBM, so that a store instruction writes whole
// 64-byte pieces (states: 16 consecutive floats of 4 instances) or one contiguous 256 bytes (controls) instead of 64
// scattered 16-byte / 4-byte ones (with every wavefront of a launch finishing together the scattered form took 50 us of a
// 300 us launch: 26 MB at 0.5 TB/s).  `so`: the wavefront's staging, at least max(64 x 17, 64 x (EU | 1)) floats; xo / uo:
// the arrays at the wavefront's first instance; mask: the lanes whose instance is stored; getx(e) / getu(e): element e of
// this lane's instance (compile-time e: the callers' trajectories are register arrays).
// (the state part and the input part on their own: the workspace arrays go the same way, five of them)
template <int EX, bool FULL, class FX>
__device__ __forceinline__ void store_wave_x(float *so, float *__restrict__ xo, int lane, unsigned long long mask, FX &&getx) {
    const int sub = lane >> 4, off = lane & 15;
    constexpr int NCH = (EX + 15) / 16;
    sfor<0, NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        sfor<0, 16>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if constexpr (c * 16 + j < EX) so[lane * 17 + j] = getx(std::integral_constant<int, c * 16 + j>{});
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = so[(4 * j + sub) * 17 + off];   // piece j: instances 4 j .. 4 j + 3, 16 floats each
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int inst = 4 * j + sub;
            const bool ok = (EX % 16 == 0 || c * 16 + off < EX) && (FULL || ((mask >> inst) & 1ull));
            if (ok) xo[inst * EX + c * 16 + off] = v[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    });
}
template <int EU, bool FULL, class FU>
__device__ __forceinline__ void store_wave_u(float *so, float *__restrict__ uo, int lane, unsigned long long mask, FU &&getu) {
    sfor<0, EU>([&](auto ee) {
        constexpr int e = decltype(ee)::value;
        so[lane * (EU | 1) + e] = getu(std::integral_constant<int, e>{});
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float w[EU];
#pragma unroll
    for (int j = 0; j < EU; ++j) {                             // flat element f of the wavefront's 64 x EU controls
        const int f = j * 64 + lane;
        w[j] = so[(f / EU) * (EU | 1) + f % EU];
    }
#pragma unroll
    for (int j = 0; j < EU; ++j) {
        const int f = j * 64 + lane;
        if (FULL || ((mask >> (f / EU)) & 1ull)) uo[f] = w[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int EX, int EU, class FX, class FU>
__device__ __forceinline__ void store_wave_coalesced(float *so, float *__restrict__ xo, float *__restrict__ uo, int lane,
                                                     unsigned long long mask, FX &&getx, FU &&getu) {
    if (mask == ~0ull) {   // every instance of the wavefront stores (the usual case): no predicates
        store_wave_x<EX, true>(so, xo, lane, mask, getx);
        store_wave_u<EU, true>(so, uo, lane, mask, getu);
    } else {
        store_wave_x<EX, false>(so, xo, lane, mask, getx);
        store_wave_u<EU, false>(so, uo, lane, mask, getu);
    }
}
// ... and the way in: a wavefront's instances' arrays from HBM — a load instruction reads whole 64-byte pieces / contiguous
// 256 bytes — through LDS into the registers of the lane that owns the instance (the kept workspace at the start of a solve:
// read in per-lane strides it cost as much again as the strided stores).  setx(e, value) / setu(e, value): element e of this
// lane's instance; lanes outside `mask` receive zeros.
template <int EX, class FX>
__device__ __forceinline__ void load_wave_x(float *so, const float *__restrict__ xi, int lane, unsigned long long mask, FX &&setx) {
    const int sub = lane >> 4, off = lane & 15;
    constexpr int NCH = (EX + 15) / 16;
    sfor<0, NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int inst = 4 * j + sub;
            const bool ok = (EX % 16 == 0 || c * 16 + off < EX) && ((mask >> inst) & 1ull);
            v[j] = ok ? xi[inst * EX + c * 16 + off] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) so[(4 * j + sub) * 17 + off] = v[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        sfor<0, 16>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if constexpr (c * 16 + j < EX) setx(std::integral_constant<int, c * 16 + j>{}, so[lane * 17 + j]);
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    });
}
template <int EU, class FU>
__device__ __forceinline__ void load_wave_u(float *so, const float *__restrict__ ui, int lane, unsigned long long mask, FU &&setu) {
    float w[EU];
#pragma unroll
    for (int j = 0; j < EU; ++j) {
        const int f = j * 64 + lane;
        w[j] = ((mask >> (f / EU)) & 1ull) ? ui[f] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < EU; ++j) {
        const int f = j * 64 + lane;
        so[(f / EU) * (EU | 1) + f % EU] = w[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    sfor<0, EU>([&](auto ee) {
        constexpr int e = decltype(ee)::value;
        setu(std::integral_constant<int, e>{}, so[lane * (EU | 1) + e]);
    });
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
constexpr int wave_stage_floats(int EU) { return 64 * 17 > 64 * (EU | 1) ? 64 * 17 : 64 * (EU | 1); }

__device__ __forceinline__ float tfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double tfma(double a, double b, double c) { return fma(a, b, c); }

// acc[m] += sum_{s < SL} sum_{k < SK} C[m*CW + s*SK + k] * (src[k] of quad lane s)
// Dot products longer than 6 terms are accumulated in two independent chains (source lanes 0,1 and
// 2,3) that are added at the end: the dependent-FMA chain is what a lone wavefront waits on.
// acc += C src (NEG: acc -= C src, the sign folded into the FMAs)
template <int G, int ROWS, int SL, int SK, int CW, bool NEG = false, class RT, class CP>
__device__ __forceinline__ void quad_matvec(RT (&acc)[ROWS], const CP C, const RT (&src)[SK]) {
    constexpr bool SPLIT = (SL * SK > 6) && SL >= 3;
    RT acc2[ROWS];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) acc2[m] = (RT)0;
    sfor<0, SL>([&](auto s) {
        constexpr int S = decltype(s)::value;
#pragma unroll
        for (int k = 0; k < SK; ++k) {
            const RT xv = gbcast<G, S>(src[k]);
#pragma unroll
            for (int m = 0; m < ROWS; ++m) {
                const RT c = NEG ? -(RT)C[m * CW + S * SK + k] : (RT)C[m * CW + S * SK + k];
                if constexpr (SPLIT && S >= 2)
                    acc2[m] = tfma(c, xv, acc2[m]);
                else
                    acc[m] = tfma(c, xv, acc[m]);
            }
        }
        if constexpr (!std::is_pointer<CP>::value && TMPC_FENCE_LDS_MATVEC == 2) __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (SPLIT) {
#pragma unroll
        for (int m = 0; m < ROWS; ++m) acc[m] += acc2[m];
    }
    // Coefficients streamed from LDS: keep the scheduler from hoisting the NEXT product's loads above
    // this one — at one wavefront per SIMD it front-loads every ds_read of a knot (hundreds of live
    // VGPRs) and the kernel spills to scratch, i.e. to HBM.
    if constexpr (!std::is_pointer<CP>::value && TMPC_FENCE_LDS_MATVEC == 1) __builtin_amdgcn_sched_barrier(0);
}

template <class T>
__device__ __forceinline__ void upmax_abs_q(T &m, T v) {
    v = v < (T)0 ? -v : v;
    m = v > m ? v : m;
}
// max over the G lanes of a group, any arithmetic type (the adaptive-rho norms are kept in the kernel's RT)
template <int G, class T>
__device__ __forceinline__ T group_max_q(T v) {
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

// OS ("one shot"): cold start and no workspace kept (the benchmark configs).  vnew / znew then overwrite
// v / z in place — the separate copies only exist to reproduce what the reference leaves in its
// workspace after a converged exit (admm.cpp:181-197) — which removes E_x + E_u floats of state per
// instance.  Same arithmetic, same results.
// UNI ("uniform"): additionally no instance can converge (fixed-iteration solves): see the loop body.
// ADP: adaptive rho (admm.cpp:147-174 with rho_benchmark.cpp:44-213), for shapes whose coefficient rows are per-lane
// registers (QuadShape::ADP_OK).  Every instance carries its own rho, Kinf and Pinf: the Kinf / Kinf^T / Pinf^T rows of the
// lane's pack are loaded from the solver's adaptive state at entry and updated in place — with that state — on the
// iterations that adapt (i > 0, i % 5 == 0), from norms gathered during that iteration's forward sweep (the stream
// kernel's ADP variant does the same with its rows in HBM columns: admm_streamg.hip.h has the derivation).
template <class S, int REFS, class RT, bool XB, bool OS, bool UNI, bool ADP = false>
__global__ __launch_bounds__(256) void admm_quad_kernel(const AdmmParams P) {
    static_assert(!UNI || OS, "the uniform variant is only built for one-shot solves");
    static_assert(!ADP || (S::ADP_OK && !UNI), "adaptive rho: per-lane coefficient registers, per-lane guard");
    static_assert(!(ADP && S::G == 1) || (REFS != REF_PER_INSTANCE && sizeof(RT) == 8),
                  "adaptive rho, one lane per instance: zero or shared references, fp64 recurrences");
    constexpr int NX = S::NX, NU = S::NU, N = S::N, G = S::G;
    constexpr int RX = S::RX, RU = S::RU, NXP = S::NXP, NUP = S::NUP;
    constexpr int NXL = S::NXL, NUL = S::NUL;
    constexpr int EX = NX * N, EU = NU * (N - 1);
    constexpr bool COEF_LDS = S::template coef_in_lds<RT, REFS>();
    // One-shot solve without an active state bound: vnew = x + g is never clamped, so g += x - vnew leaves the
    // state dual at its cold-start value, zero, for the whole solve — it is neither stored nor computed (the
    // results are bit-identical: x + 0 and (0 + x) - x are exact).
    constexpr bool NOG = OS && !XB;
    constexpr auto PL = S::template place<RT, REFS, OS, NOG>();
    // one lane per instance and LDS free: solution and workspace cross a wavefront's LDS staging on their way to / from HBM
    constexpr bool CO_STORE = G == 1 && PL.lds_floats == 0;
    __shared__ float s_stage[CO_STORE ? 4 : 1][CO_STORE ? wave_stage_floats(S::NU * (S::N - 1)) : 1];
    constexpr int STATE_LEN = PL.lds_floats > 0 ? PL.lds_floats * S::THREADS : 1;
    constexpr int T = S::THREADS;
    constexpr bool UREP = S::UREP;
    // two copies of the forward sweep (with / without the residual maxima) only where the code stays small
    constexpr bool DUAL_FWD = (RX + RU) * N <= 64;
    // -d and -Kinf^T r folded into the accumulators' starting values and FMA signs: fewer instructions where the
    // state is in registers (one lane per instance: cartpole 0.358 -> 0.352 ms); with LDS-resident state the earlier
    // operand loads cost more than they save (quadrotor 11.7 -> 12.0 ms), so quads keep the separate form unless the
    // shape asks for it (QuadShape::LOOPV bit 2: rocket N=50 with fp64 recurrences 5.88 -> 5.54 ms)
    constexpr bool FOLD = G == 1 || ((S::LOOPV & 4) != 0 && sizeof(RT) == 8);

    __shared__ float s_bnd[S::BOUNDS_LEN];
    __shared__ float s_ref[REFS == REF_SHARED ? S::REFS_LEN : 1];
    __shared__ RT s_coef[COEF_LDS ? G * S::CP : 1];
    __shared__ __align__(16) float s_state[STATE_LEN];
    static_assert(sizeof(float) * (S::BOUNDS_LEN + (REFS == REF_SHARED ? S::REFS_LEN : 1) + STATE_LEN) +
                          sizeof(RT) * (COEF_LDS ? G * S::CP : 1) <=
                      160 * 1024,
                  "workgroup LDS budget exceeded");

    const int tid = threadIdx.x;
    const RT *__restrict__ gcoef = reinterpret_cast<const RT *>(P.coef);
    for (int i = tid; i < S::BOUNDS_LEN; i += T) s_bnd[i] = P.bounds[i];
    if constexpr (COEF_LDS)  // role-major in HBM -> role-interleaved 16-byte chunks in LDS
        for (int i = tid; i < G * S::CP; i += T) s_coef[CoefLds<RT, G>::slot(i % S::CP, i / S::CP)] = gcoef[i];
    if constexpr (REFS == REF_SHARED) {
        // pack [N][G][xref[RX] uref[RU]] from knot-major xref [N][nx], uref [N-1][nu]
        for (int i = tid; i < S::REFS_LEN; i += T) {
            const int k = i / (G * S::RW), rem = i % (G * S::RW);
            const int qq = rem / S::RW, j = rem % S::RW;
            float val = 0.f;
            if (j < RX) {
                const int row = qq * RX + j;
                if (row < NX) val = P.xref[k * NX + row];
            } else {
                const int row = UREP ? (j - RX) : qq * RU + (j - RX);
                if (row < NU && k < N - 1) val = P.uref[k * NU + row];
            }
            s_ref[i] = val;
        }
    }
    __syncthreads();

    const int q = tid & (G - 1);
    const long slot = (long)blockIdx.x * S::INST_PER_BLOCK + tid / G;
    const bool active = slot < P.batch;
    const long b = (active && P.idx) ? P.idx[slot] : slot;  // instance this lane group works on
    const float *lb = s_bnd + q * S::BW;
    const float *lr = s_ref + q * S::RW;
    const float *ld = s_bnd + N * G * S::BW + q * S::DW;  // diag(Q)+rho, diag(R)+rho

    // ---- per-lane coefficient rows: VGPRs, LDS (big shapes) or scalar loads (G = 1) ----
    RT rcoef[(COEF_LDS || G == 1) ? 1 : S::CP];
    using CPtr = std::conditional_t<COEF_LDS, CoefLds<RT, G>, const RT *>;
    CPtr cbase;
    if constexpr (COEF_LDS) {
        cbase = CoefLds<RT, G>{s_coef + q * CoefLds<RT, G>::VEC};
    } else if constexpr (G == 1) {
        cbase = gcoef;  // only the sweeps' scalar blocks read it
    } else {
        const RT *cp = gcoef + q * S::CP;
#pragma unroll
        for (int i = 0; i < S::CP; ++i) rcoef[i] = cp[i];
        cbase = rcoef;
    }
    const CPtr cA = cbase + S::O_A, cAT = cbase + S::O_AT, cK = cbase + S::O_K, cB = cbase + S::O_B,
               cBT = cbase + S::O_BT, cKT = cbase + S::O_KT, cQI = cbase + S::O_QI, cPT = cbase + S::O_PT;
    float cQD[RX], cRD[RU];
#pragma unroll
    for (int m = 0; m < RX; ++m) cQD[m] = ld[m];
#pragma unroll
    for (int m = 0; m < RU; ++m) cRD[m] = ld[RX + m];
    float rho = P.rho;         // (ADP: this instance's own, re-predicted every 5th iteration)
    float rho_lin = rho;       // the rho the linear cost of the current iteration is formed with (admm.cpp:139 precedes :147)
    double rho_d = (double)P.rho;
    bool adapt_now = false;    // this iteration's forward sweep gathers the norms
    RT a_pri = 0, a_axm = 0, a_zm = 0, a_dres = 0, a_pxm = 0, a_atym = 0, a_qm = 0;
    RT accP[RX];               // ADP: Pinf' xref_{N-1} with the Pinf the linear cost was formed with
#pragma unroll
    for (int m = 0; m < RX; ++m) accP[m] = (RT)0;
    // the family's Kinf^T rows: A' g = AmBKt g + Kinf0' (B' g) — the pack has no A^T block, and AmBKt stays the family's
    RT kt0[(ADP && G != 1) ? RX * NUP : 1];
    constexpr bool ADP1 = ADP && G == 1;
    RT dK1[ADP1 ? NU : 1][ADP1 ? NX : 1];   // ADP, one lane per instance: Kinf_b - Kinf_family
    double rho_entry = 0.0;
    if constexpr (ADP1) {
#pragma unroll
        for (int a = 0; a < NU; ++a)
#pragma unroll
            for (int j = 0; j < NX; ++j) dK1[a][j] = (RT)0;
        if (active) {
            rho_d = P.adapt[b];
            rho = (float)rho_d;
#pragma unroll
            for (int a = 0; a < NU; ++a)
#pragma unroll
                for (int j = 0; j < NX; ++j) dK1[a][j] = (RT)((rho_d - P.rho_family) * P.sens[a + j * NU]);
        }
        rho_entry = rho_d;
    }
    // ADP, one lane per instance, shared references: Pinf_family' xref_{N-1} and dPinf' xref_{N-1} (wave-uniform)
    RT accA0[(ADP1 && REFS == REF_SHARED) ? NX : 1], accA1[(ADP1 && REFS == REF_SHARED) ? NX : 1];
    if constexpr (ADP1 && REFS == REF_SHARED) {
        const double *sP = P.sens + NU * NX;
#pragma unroll
        for (int m = 0; m < NX; ++m) {
            RT a0 = (RT)0, a1 = (RT)0;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const RT xrj = (RT)lr[(N - 1) * G * S::RW + j];
                a0 = tfma((RT)gcoef[S::O_PT + m * NXP + j], xrj, a0);
                a1 = tfma((RT)sP[j + m * NX], xrj, a1);
            }
            accA0[m] = a0, accA1[m] = a1;
        }
    }
    if constexpr (ADP && G != 1) {
#pragma unroll
        for (int i = 0; i < RX * NUP; ++i) kt0[i] = rcoef[S::O_KT + i];
        if (active) {   // this instance's rows from the solver's adaptive state [1 + nu nx + nx nx][batch] (rho | Kinf | Pinf)
            const long AB = P.adapt_stride;
            const double *ad = P.adapt + b;
            rho_d = ad[0];
            rho = (float)rho_d;
#pragma unroll
            for (int m = 0; m < RU; ++m) {
                const int a = UREP ? m : q * RU + m;
#pragma unroll
                for (int j = 0; j < NX; ++j)
                    if (a < NU) rcoef[S::O_K + m * NXP + j] = (RT)ad[(long)(1 + a + j * NU) * AB];
            }
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int r = q * RX + m;
                if (r < NX) {
#pragma unroll
                    for (int a = 0; a < NU; ++a) rcoef[S::O_KT + m * NUP + a] = (RT)ad[(long)(1 + a + r * NU) * AB];
#pragma unroll
                    for (int j = 0; j < NX; ++j) rcoef[S::O_PT + m * NXP + j] = (RT)ad[(long)(1 + NU * NX + j + r * NX) * AB];
                }
            }
        }
    }

    // ---- per-instance state ----
    // One accessor per trajectory: a statically indexed register array, or a per-thread column of
    // the LDS state block, as the placement says.
#define TMPC_STATE_ARRAY(name, ID, KN, RW_)                                                      \
    float name##_reg[PL.lds[S::ID] ? 1 : (KN)][PL.lds[S::ID] ? 1 : (RW_)];                      \
    float *const name##_lds = s_state + PL.off[S::ID] * T + (TMPC_LDS_ROWS ? tid * (RW_) : tid); \
    auto name##_get = [&](int k, int m) -> float {                                              \
        if constexpr (PL.lds[S::ID])                                                            \
            return TMPC_LDS_ROWS ? name##_lds[k * T * (RW_) + m] : name##_lds[(k * (RW_) + m) * T]; \
        else return name##_reg[k][m];                                                           \
    };                                                                                          \
    auto name##_set = [&](int k, int m, float val) {                                            \
        if constexpr (PL.lds[S::ID]) {                                                          \
            if (TMPC_LDS_ROWS) name##_lds[k * T * (RW_) + m] = val;                             \
            else name##_lds[(k * (RW_) + m) * T] = val;                                         \
        } else name##_reg[k][m] = val;                                                          \
    };
    TMPC_STATE_ARRAY(gx, A_G, NOG ? 1 : N, RX)   // state dual (no storage when NOG)
    auto g_get = [&](int k, int m) -> float {
        if constexpr (NOG) return 0.f;
        else return gx_get(k, m);
    };
    auto g_set = [&](int k, int m, float val) {
        if constexpr (!NOG) gx_set(k, m, val);
    };
    TMPC_STATE_ARRAY(v, A_V, N, RX)        // v (previous slack)
    TMPC_STATE_ARRAY(y, A_Y, N - 1, RU)    // input dual
    TMPC_STATE_ARRAY(z, A_Z, N - 1, RU)    // z (previous slack)
    TMPC_STATE_ARRAY(d, A_D, N - 1, RU)    // feed-forward
    TMPC_STATE_ARRAY(wx, A_W, OS ? 1 : N, RX)        // vnew  (own storage unless OS)
    TMPC_STATE_ARRAY(zx, A_ZW, OS ? 1 : N - 1, RU)   // znew  (own storage unless OS)
#undef TMPC_STATE_ARRAY
    auto w_get = [&](int k, int m) -> float {
        if constexpr (OS) return v_get(k, m);
        else return wx_get(k, m);
    };
    auto w_set = [&](int k, int m, float val) {
        if constexpr (OS) v_set(k, m, val);
        else wx_set(k, m, val);
    };
    auto zw_get = [&](int k, int m) -> float {
        if constexpr (OS) return z_get(k, m);
        else return zx_get(k, m);
    };
    auto zw_set = [&](int k, int m, float val) {
        if constexpr (OS) z_set(k, m, val);
        else zx_set(k, m, val);
    };
    RT x0[RX];  // plant state; RT so a fused closed loop does not round it to fp32 every step
    float xr[REFS == REF_PER_INSTANCE ? N : 1][RX];
    float ur[REFS == REF_PER_INSTANCE ? N - 1 : 1][RU];

#pragma unroll
    for (int m = 0; m < RX; ++m) {
        const int row = q * RX + m;
        x0[m] = (active && row < NX) ? (RT)P.x0[b * NX + row] : (RT)0;
    }
    const bool warm = active && !P.cold_start;
    // cold start = the zero workspace tiny_setup leaves (tiny_api.cpp:73-88)
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int m = 0; m < RX; ++m) {
            g_set(k, m, 0.f);
            w_set(k, m, 0.f);
            v_set(k, m, 0.f);
            if constexpr (REFS == REF_PER_INSTANCE) xr[k][m] = 0.f;
        }
#pragma unroll
    for (int k = 0; k < N - 1; ++k)
#pragma unroll
        for (int m = 0; m < RU; ++m) {
            y_set(k, m, 0.f);
            d_set(k, m, 0.f);
            zw_set(k, m, 0.f);
            z_set(k, m, 0.f);
            if constexpr (REFS == REF_PER_INSTANCE) ur[k][m] = 0.f;
        }
    bool warm_loaded = false;
    if constexpr (CO_STORE && !OS) {   // (the one-shot variants are never launched on a kept workspace: quad_entry.hip.h)
        if (!P.cold_start && !P.idx) {   // (uniform: the whole wavefront takes this path)
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(active);
            const long w0 = (long)blockIdx.x * S::INST_PER_BLOCK + (tid & ~63);
            float *so = s_stage[tid >> 6];
            if (mask) {
                load_wave_x<EX>(so, P.sg + w0 * EX, tid & 63, mask, [&](auto ee, float val) { constexpr int e = decltype(ee)::value; g_set(e / NX, e % NX, val); });
                load_wave_x<EX>(so, P.sv + w0 * EX, tid & 63, mask, [&](auto ee, float val) { constexpr int e = decltype(ee)::value; v_set(e / NX, e % NX, val); });
                load_wave_u<EU>(so, P.sy + w0 * EU, tid & 63, mask, [&](auto ee, float val) { constexpr int e = decltype(ee)::value; y_set(e / NU, e % NU, val); });
                load_wave_u<EU>(so, P.sz + w0 * EU, tid & 63, mask, [&](auto ee, float val) { constexpr int e = decltype(ee)::value; z_set(e / NU, e % NU, val); });
                load_wave_u<EU>(so, P.sd + w0 * EU, tid & 63, mask, [&](auto ee, float val) { constexpr int e = decltype(ee)::value; d_set(e / NU, e % NU, val); });
            }
            warm_loaded = true;
        }
    }
    if (warm && !warm_loaded) {
#pragma unroll
        for (int m = 0; m < RX; ++m) {
            const int row = q * RX + m;
            if (row < NX) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    g_set(k, m, P.sg[b * EX + k * NX + row]);
                    v_set(k, m, P.sv[b * EX + k * NX + row]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < RU; ++m) {
            const int row = UREP ? m : q * RU + m;
            if (row < NU) {
#pragma unroll
                for (int k = 0; k < N - 1; ++k) {
                    y_set(k, m, P.sy[b * EU + k * NU + row]);
                    z_set(k, m, P.sz[b * EU + k * NU + row]);
                    d_set(k, m, P.sd[b * EU + k * NU + row]);
                }
            }
        }
    }
    if constexpr (REFS == REF_PER_INSTANCE) {
        if (active) {
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                if (row < NX) {
#pragma unroll
                    for (int k = 0; k < N; ++k) xr[k][m] = P.xref[b * EX + k * NX + row];
                }
            }
#pragma unroll
            for (int m = 0; m < RU; ++m) {
                const int row = UREP ? m : q * RU + m;
                if (row < NU) {
#pragma unroll
                    for (int k = 0; k < N - 1; ++k) ur[k][m] = P.uref[b * EU + k * NU + row];
                }
            }
        }
    }

    auto ref_x = [&](auto kk, int m) -> float {
        constexpr int K = decltype(kk)::value;
        if constexpr (REFS == REF_SHARED) return lr[K * G * S::RW + m];
        else if constexpr (REFS == REF_PER_INSTANCE) return xr[K][m];
        else return 0.f;
    };
    auto ref_u = [&](auto kk, int m) -> float {
        constexpr int K = decltype(kk)::value;
        if constexpr (REFS == REF_SHARED) return lr[K * G * S::RW + RX + m];
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

    // Residual maxima are only needed on iterations whose termination check can matter: every
    // check when the tolerances are positive, otherwise (a residual >= 0 is never < tol <= 0, so no
    // instance can converge) only the last check, whose values the reference would report.
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;
    float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;

    // ================= fused forward sweep =================
    // forward_pass (admm.cpp:25-35) + update_slack (:43-59) + update_dual (:65-69)
    // + (RES) the residual maxima of termination_condition (:93-96), knot by knot.
    auto forward_body = [&](auto res_tag, const auto cA, const auto cK, const auto cB) {
        constexpr bool RES = decltype(res_tag)::value;
        RT x[RX];
#pragma unroll
        for (int m = 0; m < RX; ++m) x[m] = x0[m];
        if constexpr (RES) pri_x = dua_x = pri_u = dua_u = 0.f;
        RT a_gn[RX], a_xf[RX], a_vn[RX], a_xp[RX], a_gp[RX], a_up[RU], a_yp[RU];   // ADP: this knot's / the previous knot's values
#pragma unroll
        for (int m = 0; m < RX; ++m) a_gn[m] = a_xf[m] = a_vn[m] = a_xp[m] = a_gp[m] = (RT)0;
#pragma unroll
        for (int m = 0; m < RU; ++m) a_up[m] = a_yp[m] = (RT)0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            // LDS-resident constants (bounds, shared references, big coefficient packs) are re-read
            // at every knot instead of being hoisted into registers for the whole solve.
            asm volatile("" ::: "memory");
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const float xf = (float)x[m];
                const float gk = g_get(k, m);
                float vn = xf + gk;                                             // vnew = x + g
                if constexpr (XB)
                    vn = fminf(lb[k * G * S::BW + RX + m],                      // x_max.cwiseMin(
                               fmaxf(lb[k * G * S::BW + m], vn));               //   x_min.cwiseMax(vnew))
                g_set(k, m, (gk + xf) - vn);                                    // g = g + x - vnew
                if constexpr (RES) {
                    pri_x = fmaxf(pri_x, fabsf(xf - vn));
                    dua_x = fmaxf(dua_x, fabsf(v_get(k, m) - vn));
                }
                w_set(k, m, vn);
                if constexpr (ADP) a_gn[m] = (RT)((gk + xf) - vn), a_xf[m] = (RT)xf, a_vn[m] = (RT)vn;
            }
            if constexpr (ADP) {
                if (adapt_now) {   // rows of knot k - 1 that needed g_k, the terminal knot's own (see admm_streamg.hip.h)
                    if (k >= 1) {
                        RT btg[RU], atx[RX];
#pragma unroll
                        for (int m = 0; m < RU; ++m) btg[m] = (RT)0;
#pragma unroll
                        for (int m = 0; m < RX; ++m) atx[m] = (RT)0;
                        if constexpr (G == 1 && XB) {   // one lane per instance: A, B of the forward block, read transposed
#pragma unroll
                            for (int j = 0; j < NX; ++j) {
#pragma unroll
                                for (int m = 0; m < RX; ++m) atx[m] = tfma((RT)cA[j * NXP + m], a_gn[j], atx[m]);
#pragma unroll
                                for (int a = 0; a < RU; ++a) btg[a] = tfma((RT)cB[j * NUP + a], a_gn[j], btg[a]);
                            }
                        }
                        if constexpr (G != 1) {   // (one lane per instance without an active state bound: g is identically zero)
                            quad_matvec<G, RU, NXL, RX, NXP>(btg, cBT, a_gn);          // B' g_k
                            quad_matvec<G, RX, NXL, RX, NXP>(atx, cAT, a_gn);          // AmBKt g_k
                            if constexpr (UREP) {
#pragma unroll
                                for (int m = 0; m < RX; ++m) atx[m] = tfma(kt0[m * NUP], btg[0], atx[m]);   // + Kinf0' B' g_k = A' g_k
                            } else {
                                quad_matvec<G, RX, NUL, RU, NUP>(atx, (const RT *)kt0, btg);
                            }
                        }
#pragma unroll
                        for (int m = 0; m < RX; ++m) {
                            if (k >= 2) atx[m] -= a_gp[m];
                            const RT qv = (RT)cQD[m] * a_xp[m];
                            upmax_abs_q(a_dres, qv + qv + atx[m]);
                            upmax_abs_q(a_pxm, qv);
                            upmax_abs_q(a_qm, qv);
                            upmax_abs_q(a_atym, atx[m]);
                            upmax_abs_q(a_pri, a_vn[m]);   // A x + B u - x_k vanishes against the rollout's own x_k
                            upmax_abs_q(a_zm, a_vn[m]);
                        }
#pragma unroll
                        for (int m = 0; m < RU; ++m) {
                            const RT px = (RT)cRD[m] * a_up[m], aty = a_yp[m] + btg[m];
                            upmax_abs_q(a_dres, px + px + aty);
                            upmax_abs_q(a_pxm, px);
                            upmax_abs_q(a_qm, px);
                            upmax_abs_q(a_atym, aty);
                        }
                    }
                    if (k == N - 1) {
                        RT px[RX];
#pragma unroll
                        for (int m = 0; m < RX; ++m) px[m] = (RT)0;
                        if constexpr (G == 1) {   // Pinf_b = Pinf_family + (rho_b - rho_family) dPinf/drho
                            const SBlock<RT, S::PT_LEN / 8> pt(gcoef + S::O_PT);
                            quad_matvec<G, RX, NXL, RX, NXP>(px, pt.at(0), a_xf);
                            const double dr = rho_d - P.rho_family, *sP = P.sens + NU * NX;
#pragma unroll
                            for (int m = 0; m < RX; ++m) {
                                RT t2 = (RT)0;
#pragma unroll
                                for (int j = 0; j < NX; ++j) t2 = tfma((RT)sP[j + m * NX], a_xf[j], t2);
                                px[m] = tfma((RT)dr, t2, px[m]);
                            }
                        } else {
                            quad_matvec<G, RX, NXL, RX, NXP>(px, cPT, a_xf);
                        }
#pragma unroll
                        for (int m = 0; m < RX; ++m) {
                            const RT qv = (RT)cQD[m] * a_xf[m], aty = (k >= 1) ? -a_gn[m] : (RT)0;
                            upmax_abs_q(a_dres, px[m] + qv + aty);
                            upmax_abs_q(a_pxm, px[m]);
                            upmax_abs_q(a_qm, qv);
                            upmax_abs_q(a_atym, aty);
                        }
                    }
#pragma unroll
                    for (int m = 0; m < RX; ++m) a_xp[m] = a_xf[m], a_gp[m] = a_gn[m];
                }
            }
            if (k < N - 1) {
                RT u[RU], xn[RX];
#pragma unroll
                for (int m = 0; m < RU; ++m) u[m] = FOLD ? -(RT)d_get(k, m) : (RT)0;
#pragma unroll
                for (int m = 0; m < RX; ++m) xn[m] = (RT)0;
                quad_matvec<G, RU, NXL, RX, NXP, FOLD>(u, cK, x);                  // Kinf x  (FOLD: u = -d - Kinf x)
                quad_matvec<G, RX, NXL, RX, NXP>(xn, cA, x);                       // A x
#pragma unroll
                for (int m = 0; m < RU; ++m) {
                    if constexpr (!FOLD) u[m] = -u[m] - (RT)d_get(k, m);        // u = -Kinf x - d
                    if constexpr (ADP1) {                                       // ... with the instance's own Kinf
#pragma unroll
                        for (int j = 0; j < NX; ++j) u[m] = tfma(-dK1[m][j], x[j], u[m]);
                    }
                    const float uf = (float)u[m];
                    const float yk = y_get(k, m);
                    float zn = uf + yk;                                         // znew = u + y
                    zn = fminf(lb[k * G * S::BW + 2 * RX + RU + m],
                               fmaxf(lb[k * G * S::BW + 2 * RX + m], zn));
                    y_set(k, m, (yk + uf) - zn);                                // y = y + u - znew
                    if constexpr (RES) {
                        pri_u = fmaxf(pri_u, fabsf(uf - zn));
                        dua_u = fmaxf(dua_u, fabsf(z_get(k, m) - zn));
                    }
                    zw_set(k, m, zn);
                    if constexpr (ADP) {
                        if (adapt_now) {
                            upmax_abs_q(a_pri, (RT)uf - (RT)zn);
                            upmax_abs_q(a_axm, (RT)uf);
                            upmax_abs_q(a_zm, (RT)zn);
                            a_up[m] = (RT)uf, a_yp[m] = (RT)((yk + uf) - zn);
                        }
                    }
                }
                // A x does not wait for u: both mat-vecs of x issue side by side, B u joins last
                if constexpr (UREP) {
#pragma unroll
                    for (int m = 0; m < RX; ++m) xn[m] = tfma((RT)cB[m * NUP], u[0], xn[m]);  // + B u
                } else {
                    quad_matvec<G, RX, NUL, RU, NUP>(xn, cB, u);                   // + B u
                }
#pragma unroll
                for (int m = 0; m < RX; ++m) x[m] = xn[m];
            }
        }
    };

    // Closed loop (examples/cartpole_example_mpc.jl:35-51): mpc_steps > 1 repeats
    //   solve -> u0 = controls[:,0] -> x0 = A x0 + B u0 -> set_x0
    // inside the launch, the warm-start workspace never leaving the registers.
    const int n_steps = P.mpc_steps > 1 ? P.mpc_steps : 1;
    for (int step = 0; step < n_steps; ++step) {
    if (step > 0) {  // solve() entry: only the counters are reset (admm.cpp:112-115)
        it = 0;
        conv = 0;
        ct_count = ct;
    }
    auto backward_body = [&](const auto cAT, const auto cBT, const auto cKT, const auto cQI, const auto cPT) {
            // ================= fused backward sweep =================
            // v = vnew, z = znew (admm.cpp:196-197); update_linear_cost (:75-83) and
            // backward_pass_grad (:13-20) knot by knot, q/r/p never stored.
            RT p[RX];
            {
                RT acc[RX];
#pragma unroll
                for (int m = 0; m < RX; ++m) acc[m] = ADP ? accP[m] : (RT)0;
                if constexpr (REFS != REF_ZERO && !ADP) {
                    RT xrl[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) xrl[m] = (RT)ref_x(std::integral_constant<int, N - 1>{}, m);
                    quad_matvec<G, RX, NXL, RX, NXP>(acc, cPT, xrl);           // (Xref_{N-1}^T Pinf)^T
                }
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    const float wN = w_get(N - 1, m);
                    p[m] = -acc[m] - (RT)(rho_lin * (wN - g_get(N - 1, m)));    // admm.cpp:81-82
                    v_set(N - 1, m, wN);
                }
            }
            sfor<0, N - 1>([&](auto kk) {
                constexpr int k = N - 2 - decltype(kk)::value;
                constexpr std::integral_constant<int, k> kc{};
                asm volatile("" ::: "memory");
                RT r[RU], qk[RX];
#pragma unroll
                for (int m = 0; m < RU; ++m) {
                    float rr = 0.f;
                    if constexpr (REFS != REF_ZERO) rr = -(ref_u(kc, m) * cRD[m]);  // -(Uref .* R)
                    const float zk = zw_get(k, m);
                    r[m] = (RT)(rr - rho_lin * (zk - y_get(k, m)));         // admm.cpp:77-78
                    z_set(k, m, zk);
                }
#pragma unroll
                for (int m = 0; m < RX; ++m) {
                    float qq = 0.f;
                    if constexpr (REFS != REF_ZERO) qq = -(ref_x(kc, m) * cQD[m]);  // -(Xref .* Q)
                    const float wk = w_get(k, m);
                    qk[m] = (RT)(qq - rho_lin * (wk - g_get(k, m)));        // admm.cpp:79-80
                    v_set(k, m, wk);
                }
                RT t[RU];
#pragma unroll
                for (int m = 0; m < RU; ++m) t[m] = r[m];
                quad_matvec<G, RU, NXL, RX, NXP>(t, cBT, p);                   // B^T p_{k+1} + r_k
                RT dn[RU];
                if constexpr (UREP) {
                    dn[0] = (RT)cQI[0] * t[0];                              // d_k = Quu_inv (...)
                } else {
#pragma unroll
                    for (int m = 0; m < RU; ++m) dn[m] = (RT)0;
                    quad_matvec<G, RU, NUL, RU, NUP>(dn, cQI, t);
                }
#pragma unroll
                for (int m = 0; m < RU; ++m) d_set(k, m, (float)dn[m]);
                // p_k = q_k + AmBKt p_{k+1} - Kinf^T r_k (admm.cpp:18)
                RT ap[RX];
                if constexpr (FOLD) {  // the part that does not wait for p_{k+1} first, its sign folded into the FMAs
                    if constexpr (UREP) {
#pragma unroll
                        for (int m = 0; m < RX; ++m) ap[m] = tfma(-(RT)cKT[m * NUP], r[0], qk[m]);
                    } else {
#pragma unroll
                        for (int m = 0; m < RX; ++m) ap[m] = qk[m];
                        quad_matvec<G, RX, NUL, RU, NUP, true>(ap, cKT, r);
                    }
                    if constexpr (ADP1) {                                       // - dK' r: the instance's own Kinf in - Kinf' r
#pragma unroll
                        for (int m = 0; m < RX; ++m)
#pragma unroll
                            for (int a = 0; a < NU; ++a) ap[m] = tfma(-dK1[a][m], r[a], ap[m]);
                    }
                    quad_matvec<G, RX, NXL, RX, NXP>(ap, cAT, p);
#pragma unroll
                    for (int m = 0; m < RX; ++m) p[m] = ap[m];
                } else {
                    RT kr[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) ap[m] = qk[m];
                    quad_matvec<G, RX, NXL, RX, NXP>(ap, cAT, p);              // q_k + AmBKt p_{k+1}
                    if constexpr (UREP) {
#pragma unroll
                        for (int m = 0; m < RX; ++m) kr[m] = (RT)cKT[m * NUP] * r[0];  // Kinf^T r_k
                    } else {
#pragma unroll
                        for (int m = 0; m < RX; ++m) kr[m] = (RT)0;
                        quad_matvec<G, RX, NUL, RU, NUP>(kr, cKT, r);
                    }
#pragma unroll
                    for (int m = 0; m < RX; ++m) p[m] = ap[m] - kr[m];
                }
            });
    };
    auto backward_sweep = [&]() {
        if constexpr (G == 1) {
            const SBlock<RT, S::BWD_LEN / 8> blk(gcoef + S::O_AT);
            if constexpr (REFS != REF_ZERO) {
                const SBlock<RT, S::PT_LEN / 8> pt(gcoef + S::O_PT);
                backward_body(blk.at(0), blk.at(S::O_BT - S::O_AT), blk.at(S::O_KT - S::O_AT),
                              blk.at(S::O_QI - S::O_AT), pt.at(0));
            } else {
                backward_body(blk.at(0), blk.at(S::O_BT - S::O_AT), blk.at(S::O_KT - S::O_AT),
                              blk.at(S::O_QI - S::O_AT), blk.at(0));  // Pinf^T unused without references
            }
        } else {
            backward_body(cAT, cBT, cKT, cQI, cPT);
        }
    };

    auto forward_sweep = [&](auto res_tag) {
        if constexpr (G == 1) {
            const SBlock<RT, S::FWD_LEN / 8> blk(gcoef + S::O_A);
            forward_body(res_tag, blk.at(S::O_A), blk.at(S::O_K), blk.at(S::O_B));
        } else {
            forward_body(res_tag, cA, cK, cB);
        }
    };

    for (int i = 0; i < P.max_iter; ++i) {
        // One ADMM iteration of this lane's instance.
        auto iterate = [&]() {
            bool check = false;
            if (ct > 0) {
                if (--ct_count == 0) {
                    check = true;
                    ct_count = ct;
                }
            }
            const bool need_res = check && (can_converge || it + 1 == last_check_it);
            rho_lin = rho;
            if constexpr (ADP) {
                adapt_now = i > 0 && i % 5 == 0;                                // admm.cpp:147 (loop index, before it is bumped)
                a_pri = a_axm = a_zm = a_dres = a_pxm = a_atym = a_qm = (RT)0;
                if constexpr (ADP1 && REFS == REF_SHARED) {
                    const RT dr = (RT)(rho_d - P.rho_family);
#pragma unroll
                    for (int m = 0; m < RX; ++m) accP[m] = tfma(dr, accA1[m], accA0[m]);
                }
                if constexpr (REFS != REF_ZERO && G != 1) {                     // the terminal cost with the Pinf of this iteration's linear cost
                    RT xrl[RX];
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        xrl[m] = (RT)ref_x(std::integral_constant<int, N - 1>{}, m);
                        accP[m] = (RT)0;
                    }
                    quad_matvec<G, RX, NXL, RX, NXP>(accP, cPT, xrl);
                }
            }
            if constexpr (DUAL_FWD) {
                if (need_res)
                    forward_sweep(std::true_type{});
                else
                    forward_sweep(std::false_type{});
            } else {
                forward_sweep(std::true_type{});
            }
            it += 1;  // admm.cpp:143
            if constexpr (ADP) {
                if (adapt_now) {
                    // predict_rho (rho_benchmark.cpp:173-195), then the first-order update of Kinf, Pinf (admm.cpp:160-172)
                    const RT pri = group_max_q<G>(a_pri), axm = group_max_q<G>(a_axm), zm = group_max_q<G>(a_zm),
                             dres = group_max_q<G>(a_dres), pxm = group_max_q<G>(a_pxm), atym = group_max_q<G>(a_atym),
                             qm = group_max_q<G>(a_qm);
                    const RT eps = (RT)1e-10, prin = axm > zm ? axm : zm;
                    RT duan = pxm > atym ? pxm : atym;
                    duan = qm > duan ? qm : duan;
                    const RT ratio = (pri / (prin + eps)) / (dres / (duan + eps) + eps);
                    RT nrho = (RT)rho_d * (RT)sqrt((double)ratio);
                    if (P.rho_clip) nrho = nrho < (RT)P.rho_min ? (RT)P.rho_min : (nrho > (RT)P.rho_max ? (RT)P.rho_max : nrho);
                    const double delta = (double)nrho - rho_d;
                    if constexpr (G == 1) {   // the correction moves; the solver's state is written once, in the epilogue
#pragma unroll
                        for (int a = 0; a < NU; ++a)
#pragma unroll
                            for (int j = 0; j < NX; ++j) dK1[a][j] += (RT)(delta * P.sens[a + j * NU]);
                    }
                    const long AB = P.adapt_stride;
                    double *ad = P.adapt + b;
                    const double *sK = P.sens, *sP = P.sens + NU * NX;
                    if constexpr (G != 1) {
#pragma unroll
                    for (int m = 0; m < RU; ++m) {
                        const int a = UREP ? m : q * RU + m;
                        if (a < NU) {
#pragma unroll
                            for (int j = 0; j < NX; ++j) {   // row a of Kinf: the solver's state and this lane's register
                                const double v = ad[(long)(1 + a + j * NU) * AB] + delta * sK[a + j * NU];
                                if (!UREP || q == 0) ad[(long)(1 + a + j * NU) * AB] = v;
                                rcoef[S::O_K + m * NXP + j] = (RT)v;
                            }
                        }
                    }
#pragma unroll
                    for (int m = 0; m < RX; ++m) {
                        const int r = q * RX + m;
                        if (r < NX) {
#pragma unroll
                            for (int a = 0; a < NU; ++a)
                                rcoef[S::O_KT + m * NUP + a] = (RT)((double)rcoef[S::O_KT + m * NUP + a] + delta * sK[a + r * NU]);
#pragma unroll
                            for (int j = 0; j < NX; ++j) {   // column r of Pinf
                                const double v = ad[(long)(1 + NU * NX + j + r * NX) * AB] + delta * sP[j + r * NX];
                                ad[(long)(1 + NU * NX + j + r * NX) * AB] = v;
                                rcoef[S::O_PT + m * NXP + j] = (RT)v;
                            }
                        }
                    }
                    if (q == 0) ad[0] = (double)nrho;
                    }   // G != 1
                    (void)AB, (void)ad, (void)sK, (void)sP;
                    rho_d = (double)nrho;
                    rho = (float)nrho;
                }
            }

            // ================= termination_condition (admm.cpp:89-107) =================
            if (need_res) {
                res0 = group_max<G>(pri_x);
                res1 = group_max<G>(dua_x) * rho;
                res2 = group_max<G>(pri_u);
                res3 = group_max<G>(dua_u) * rho;
                if constexpr (!UNI) {
                    if (res0 < P.abs_pri_tol && res2 < P.abs_pri_tol && res1 < P.abs_dua_tol &&
                        res3 < P.abs_dua_tol)
                        conv = 1;
                }
            }
            if constexpr (UNI)
                backward_sweep();
            else if (!conv)
                backward_sweep();
        };
        if constexpr (UNI) {
            // No instance can converge (a tolerance is <= 0): every lane runs every iteration, so the
            // body is straight-line code for the whole wavefront.  With the per-lane guard below, each
            // loop-carried trajectory value needs a copy at the join of the divergent region — 300+
            // moves per iteration and, worse, old and new values alive together (twice the registers).
            // Padding lanes (b >= batch) iterate on zeros; their results are never stored.
            iterate();
        } else {
            if (active && !conv) iterate();
            // every instance of this wavefront finished?  (wave-uniform exit)
            if (!__builtin_amdgcn_ballot_w64(active && !conv)) break;
        }
    }
    if (P.mpc_steps > 0) {
        // apply the first control to the plant model and log the step
        RT u0[RU], xn[RX];
#pragma unroll
        for (int m = 0; m < RU; ++m) u0[m] = (RT)zw_get(0, m);
#pragma unroll
        for (int m = 0; m < RX; ++m) xn[m] = (RT)0;
        auto plant = [&](const auto pA, const auto pB) {
            quad_matvec<G, RX, NXL, RX, NXP>(xn, pA, x0);
            if constexpr (UREP) {
#pragma unroll
                for (int m = 0; m < RX; ++m) xn[m] = tfma((RT)pB[m * NUP], u0[0], xn[m]);
            } else {
                quad_matvec<G, RX, NUL, RU, NUP>(xn, pB, u0);
            }
        };
        if constexpr (G == 1) {
            const SBlock<RT, S::FWD_LEN / 8> blk(gcoef + S::O_A);
            plant(blk.at(S::O_A), blk.at(S::O_B));
        } else {
            plant(cA, cB);
        }
#pragma unroll
        for (int m = 0; m < RX; ++m) x0[m] = xn[m];
        if (active) {
            const long so = (b * n_steps + step);
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                if (row < NX) P.mpc_x[so * NX + row] = (float)x0[m];
            }
#pragma unroll
            for (int m = 0; m < RU; ++m) {
                const int row = q * RU + m;
                if (row < NU) P.mpc_u[so * NU + row] = zw_get(0, m);
            }
            if (q == 0) P.mpc_iter[so] = conv ? it : -it;  // sign carries the solved flag
        }
    }
    }  // mpc step

    // ================= epilogue: solution, status, warm-start state =================
    // solution = projected slack of the last executed iteration (admm.cpp:187-188,204-205)
    if constexpr (CO_STORE) {
        if (!P.idx) {   // (with an index list a wavefront's instances are not neighbours in the arrays)
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(active);
            const long w0 = (long)blockIdx.x * S::INST_PER_BLOCK + (tid & ~63);
            if (mask) {
                store_wave_coalesced<EX, EU>(s_stage[tid >> 6], P.xout + w0 * EX, P.uout + w0 * EU, tid & 63, mask,
                                             [&](auto ee) { constexpr int e = decltype(ee)::value; return w_get(e / NX, e % NX); },
                                             [&](auto ee) { constexpr int e = decltype(ee)::value; return zw_get(e / NU, e % NU); });
                if constexpr (!OS)
                if (P.save_state) {   // the workspace the same way (0.42 ms per 65 536 kept-workspace cartpole solves: 80 us of it were these arrays in per-lane strides)
                    float *so = s_stage[tid >> 6];
                    store_wave_coalesced<EX, EU>(so, P.sg + w0 * EX, P.sy + w0 * EU, tid & 63, mask,
                                                 [&](auto ee) { constexpr int e = decltype(ee)::value; return g_get(e / NX, e % NX); },
                                                 [&](auto ee) { constexpr int e = decltype(ee)::value; return y_get(e / NU, e % NU); });
                    store_wave_coalesced<EX, EU>(so, P.sv + w0 * EX, P.sz + w0 * EU, tid & 63, mask,
                                                 [&](auto ee) { constexpr int e = decltype(ee)::value; return v_get(e / NX, e % NX); },
                                                 [&](auto ee) { constexpr int e = decltype(ee)::value; return z_get(e / NU, e % NU); });
                    store_wave_u<EU, false>(so, P.sd + w0 * EU, tid & 63, mask,
                                            [&](auto ee) { constexpr int e = decltype(ee)::value; return d_get(e / NU, e % NU); });
                }
            }
        }
    }
    if (active) {
        if (!CO_STORE || P.idx) {
#pragma unroll
        for (int m = 0; m < RX; ++m) {
            const int row = q * RX + m;
            if (row < NX) {
#pragma unroll
                for (int k = 0; k < N; ++k) P.xout[b * EX + k * NX + row] = w_get(k, m);
            }
        }
#pragma unroll
        for (int m = 0; m < RU; ++m) {
            const int row = q * RU + m;
            if (row < NU) {
#pragma unroll
                for (int k = 0; k < N - 1; ++k) P.uout[b * EU + k * NU + row] = zw_get(k, m);
            }
        }
        }
        if (P.mpc_steps > 0) {
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                if (row < NX) P.x0_out[b * NX + row] = (float)x0[m];
            }
        }
        if constexpr (ADP1) {
            // the adaptive state as the reference leaves it (admm.cpp:160-172 accumulated): family + (rho_b - rho_family) x tables
            if (rho_d != rho_entry) {
                const double dr = rho_d - P.rho_family, *sK = P.sens, *sP = P.sens + NU * NX;
                const long AB = P.adapt_stride;
                double *ad = P.adapt + b;
                for (int a = 0; a < NU; ++a)
                    for (int j = 0; j < NX; ++j) ad[(long)(1 + a + j * NU) * AB] = (double)gcoef[S::O_K + a * NXP + j] + dr * sK[a + j * NU];
                for (int r = 0; r < NX; ++r)
                    for (int j = 0; j < NX; ++j)
                        ad[(long)(1 + NU * NX + j + r * NX) * AB] = (double)gcoef[S::O_PT + r * NXP + j] + dr * sP[j + r * NX];
                ad[0] = rho_d;
            }
        }
        if (q == 0) {
            P.iter[b] = P.iter_offset + it;
            P.solved[b] = conv;
            P.res[b * 4 + 0] = res0;
            P.res[b * 4 + 1] = res1;
            P.res[b * 4 + 2] = res2;
            P.res[b * 4 + 3] = res3;
        }
        if (P.save_state && (!CO_STORE || OS || P.idx)) {
#pragma unroll
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                if (row < NX) {
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        P.sg[b * EX + k * NX + row] = g_get(k, m);
                        P.sv[b * EX + k * NX + row] = v_get(k, m);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < RU; ++m) {
                const int row = q * RU + m;
                if (row < NU) {
#pragma unroll
                    for (int k = 0; k < N - 1; ++k) {
                        P.sy[b * EU + k * NU + row] = y_get(k, m);
                        P.sz[b * EU + k * NU + row] = z_get(k, m);
                        P.sd[b * EU + k * NU + row] = d_get(k, m);
                    }
                }
            }
        }
    }
    // global status block: wavefront max of the residuals, count of unsolved instances
    {
        float m0 = active ? res0 : 0.f, m1 = active ? res1 : 0.f, m2 = active ? res2 : 0.f,
              m3 = active ? res3 : 0.f;
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
