// Fused TinyMPC ADMM kernel for gfx950 — "lean" layout: the benchmark's calling pattern of the small shapes with as few
// vector instructions per ADMM iteration as the arithmetic allows.
//
// What it computes: the reference's solve() loop (src/codegen_src/tinympc/admm.cpp:109-207; phases :13-107) for one-shot
// solves — cold start (the zero workspace tiny_setup leaves, tiny_api.cpp:73-88), nothing of the workspace kept — of a
// box-constrained family, zero or shared references, fp64 recurrences.  One lane per instance, like quad<..., g1>
// (admm_quad.hip.h), which stays the kernel of every other calling pattern of the shape (workspace kept, closed loop,
// per-instance references, adaptive rho, fp32 recurrences).  The benchmark's pattern — no active state bound, zero
// references — is the XB = false, REFS = REF_ZERO instantiation described first; XB / REF_SHARED add what they need and
// nothing to that instantiation (profiles/r04_cartpole_isa_census.json is its loop).
//
// Why a separate kernel (round 3 review, item 1): quad<4,1,20,g1> executed 1 506 vector instructions per iteration, of
// which 911 were the recurrences' fp64 FMAs; 268 were v_accvgpr moves (its state homed in AGPRs and copied through VGPRs
// every iteration) and 228 fp32 <-> fp64 conversions.  A lone wavefront issues one vector instruction per 4 cycles
// whatever its type (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"), so the instruction count IS the time.  Here:
//   * without an active state bound the state slack is the rollout itself (vnew = x + g clamps nothing, so g stays 0 and
//     vnew = x: admm.cpp:46-58, :67-68) — the trajectory x is kept in fp64 registers and IS v: no x -> fp32 -> fp64 round
//     trip between the forward and the backward sweep (8 conversions per knot), and v is MORE accurate than the fp32 copy;
//   * the whole iterated state — x (fp64), y, znew, d (fp32): 2 nx N + 3 nu (N-1) registers = 217 for cartpole N = 20 —
//     plus the working set of a knot fits the 256 architectural VGPRs: no AGPR homes, no moves, and the kernel may run two
//     wavefronts per SIMD when the batch has them (__launch_bounds__(256, 2));
//   * the backward recursion runs on p~ = -p / rho, r~ = -r / rho: with zero references q = -rho x, r = -rho (z - y)
//     (admm.cpp:77-80), so p~_k = x_k + AmBKt p~_{k+1} - Kinf' r~_k starts from the x register itself and the products by
//     rho disappear; d_k = (-rho Quu_inv) (B' p~_{k+1} + r~_k) (admm.cpp:17-18);
//   * the rollout is regrouped as x+ = (A - B Kinf) x - B d, u = -Kinf x - d (admm.cpp:29-30; fp64, results move by
//     ~1e-16): x+ does not wait for u, and (A - B Kinf) is the transpose of the AmBKt the backward sweep reads, so ONE set
//     of 25 fp64 coefficients (50 SGPRs) serves both sweeps and stays resident for the whole solve — no per-sweep scalar
//     reloads.  (The host only selects this kernel when cache.AmBKt equals (A - B Kinf)' — set_cache_terms may break that.)
// XB (some enabled state bound is finite): the state slack is no longer the rollout.  The rollout is then a running fp64
// vector; per element the iteration keeps the state dual g and q~ = vnew - g (what the backward sweep starts from) in fp32
// — still 217 registers — and forms them in fp32 from the rounded x as every fp32-state kernel does (admm.cpp:46-58, 67-68,
// 79-80); the solution vnew = q~ + g is re-clamped at the store.  REF_SHARED: the reference terms of update_linear_cost
// (admm.cpp:77-82) scaled by -1 / rho — Q~ xref_k / rho, R~ uref_k / rho, Pinf' xref_{N-1} / rho — are formed once per
// workgroup in fp64 and read from LDS (broadcast) in the backward sweep.
// Termination (admm.cpp:89-107) as in the quad kernel: residual maxima only on the iterations whose check can matter; with
// positive tolerances (LIVE) an instance that converges stores its solution at that iteration and its lane idles on
// (what the matrix-core kernels do), the wavefront leaves when all its instances are done.
#pragma once
#include <hip/hip_runtime.h>

#include "admm_params.h"
#include "admm_quad.hip.h"   // SBlock, sfor

#ifndef TMPC_LEAN_KNOT_BARRIER
#define TMPC_LEAN_KNOT_BARRIER 0   // 1: a scheduling barrier per knot (keeps the scheduler from hoisting a later knot's loads / conversions)
#endif
#ifndef TMPC_LEAN_SPLITK
#define TMPC_LEAN_SPLITK 0    // 1: Kinf x as two chains of two (+ an add): shorter dependent chain, one more instruction
#endif

namespace tmpc {

template <int NX, int NU>
struct LeanPack {
    static constexpr LeanLayout LL = lean_layout(NX, NU);
    static constexpr int O_M = LL.oM, O_K = LL.oK, O_B = LL.oB, O_C = LL.oC, O_P = LL.oP, LEN = LL.len;
    static constexpr int NLOADS = LL.padded / 8;     // s_load_dwordx16 per 8 doubles
};

// clamp(t, lo, hi) as one v_med3_f32 (lo <= hi; +-inf for "no bound")
__device__ __forceinline__ float clamp3(float t, float lo, float hi) { return __builtin_amdgcn_fmed3f(t, lo, hi); }
__device__ __forceinline__ double clamp3(double t, double lo, double hi) { return fmin(fmax(t, lo), hi); }
__device__ __forceinline__ float lean_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double lean_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float lean_abs(float a) { return fabsf(a); }
__device__ __forceinline__ double lean_abs(double a) { return fabs(a); }

// ONE: the launch has at most one wavefront per SIMD (batch <= 256 x CUs), so the kernel may take the whole register file:
// the feed-forward term d is then kept in fp64 too (nu (N-1) more registers, two conversions per knot fewer: 3.5 % of the
// instructions).  Otherwise (fixed-iteration solves of larger batches) the kernel is held to 256 registers and two wavefronts
// share a SIMD: within 3 % of 512-register wavefronts taking turns without a state bound, 12 % better with one; the
// tolerance-terminated (LIVE) kernels spill at 256 registers and are only built in the ONE form (lean_entry.hip.h).
// ST: the type of the slack / dual state.  float: the library's precision 0 (fp64 recurrences, fp32 state).  double: the
// reference's own arithmetic end to end (types.hpp:15) — precision 2 for one-shot solves of the shapes this kernel holds, at
// this kernel's speed instead of the generic kernel's; only ever specialised on request (jit.cpp), in the ONE form.
template <int NX, int NU, int N, bool LIVE, bool UBK, bool ONE, bool XB = false, int REFS = REF_ZERO, class ST = float>
__global__ __launch_bounds__(256, (ONE ? 1 : 2)) void admm_lean_kernel(const AdmmParams P) {
    constexpr bool F64 = std::is_same<ST, double>::value;
    static_assert(!F64 || ONE, "fp64 state: the 512-register form");
    static_assert(REFS == REF_ZERO || REFS == REF_SHARED, "lean kernel: zero or shared references");
#ifdef TMPC_LEAN_CLOCK_PROBE
    const unsigned long long probe_entry = __builtin_amdgcn_s_memrealtime();
#endif
    using L = LeanPack<NX, NU>;
    constexpr int EX = NX * N, EU = NU * (N - 1);
    constexpr int BW = 2 * NX + 2 * NU;              // the quad kernel's bounds pack, one lane per instance: [N][xmin xmax umin umax]
    static_assert(L::NLOADS <= 4, "coefficient block too large for SGPRs");

    __shared__ float s_bnd[UBK ? 1 : 2 * NU * (N - 1)];
    __shared__ float s_xb[XB ? 2 * NX * N : 1];                                  // [knot][x_min[NX] x_max[NX]]
    __shared__ double s_cq[REFS == REF_SHARED ? NX * N : 1], s_cr[REFS == REF_SHARED ? NU * (N - 1) : 1], s_cpt[REFS == REF_SHARED ? NX : 1];
    const int tid = threadIdx.x;
    if constexpr (!UBK) {
        for (int i = tid; i < 2 * NU * (N - 1); i += 256) {
            const int k = i / (2 * NU), j = i % (2 * NU);
            s_bnd[i] = P.bounds[k * BW + 2 * NX + j];
        }
    }
    if constexpr (XB)
        for (int i = tid; i < 2 * NX * N; i += 256) s_xb[i] = P.bounds[(i / (2 * NX)) * BW + i % (2 * NX)];
    if constexpr (REFS == REF_SHARED) {
        // -(Xref .* Q~) / (-rho), -(Uref .* R~) / (-rho), (Xref_{N-1}' Pinf)' / rho  (admm.cpp:77-82 on the scaled recursion)
        const float *qd = P.bounds + N * BW, *rd = qd + NX;                      // diag(Q) + rho, diag(R) + rho behind the bounds
        const double irho = 1.0 / P.rho_family;
        for (int i = tid; i < NX * N; i += 256) s_cq[i] = (double)P.xref[i] * (double)qd[i % NX] * irho;
        for (int i = tid; i < NU * (N - 1); i += 256) s_cr[i] = (double)P.uref[i] * (double)rd[i % NU] * irho;
        if (tid < NX) {
            double acc = 0.0;
            for (int j = 0; j < NX; ++j) acc = fma(P.lean[L::O_P + j * NX + tid], (double)P.xref[(N - 1) * NX + j], acc);
            s_cpt[tid] = acc * irho;
        }
    }
    if constexpr (!UBK || XB || REFS == REF_SHARED) __syncthreads();
    const long b = (long)blockIdx.x * 256 + tid;     // (no index list: the solver sends compacted / chunked solves to the quad kernel)
    const bool active = b < P.batch;
    const int lane = tid & 63;
    // staging of a wavefront's solution for the final store (below): [64 instances][16 + 1 floats], or [64][nu (N-1)]
    __shared__ float s_stage[4][wave_stage_floats(EU)];

    const SBlock<double, L::NLOADS> blk(P.lean);
    const auto cM = blk.at(L::O_M), cK = blk.at(L::O_K), cB = blk.at(L::O_B), cC = blk.at(L::O_C);

    ST lo[NU], hi[NU];
#pragma unroll
    for (int a = 0; a < NU; ++a) lo[a] = (ST)P.bounds[2 * NX + a], hi[a] = (ST)P.bounds[2 * NX + NU + a];

    // ---- the iterated state: x (= v = vnew) in fp64 — or, with an active state bound, the state dual g and q~ = vnew - g in
    // fp32 beside a running x — and input dual / slack / feed-forward in fp32 ----
    double X[XB ? 1 : N][NX];           // XB: X[0] is the plant state x0 only
    ST G[XB ? N : 1][NX], QT[XB ? N : 1][NX];
    ST Y[N - 1][NU], Z[N - 1][NU];
    using DT = std::conditional_t<ONE, double, float>;
    DT D[N - 1][NU];
#pragma unroll
    for (int m = 0; m < NX; ++m) X[0][m] = active ? (double)P.x0[b * NX + m] : 0.0;
#pragma unroll
    for (int k = 1; k < (XB ? 1 : N); ++k)
#pragma unroll
        for (int m = 0; m < NX; ++m) X[k][m] = 0.0;
#pragma unroll
    for (int k = 0; k < (XB ? N : 1); ++k)
#pragma unroll
        for (int m = 0; m < NX; ++m) G[k][m] = (ST)0, QT[k][m] = (ST)0;
#pragma unroll
    for (int k = 0; k < N - 1; ++k)
#pragma unroll
        for (int a = 0; a < NU; ++a) Y[k][a] = (ST)0, Z[k][a] = (ST)0, D[k][a] = (DT)0;

#ifdef TMPC_LEAN_CLOCK_PROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    int it = 0, conv = 0;
    ST res0 = 0, res1 = 0, res2 = 0, res3 = 0;
    const int ct = P.check_termination;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;
    const ST rho = F64 ? (ST)P.rho_family : (ST)P.rho;
    const ST ptol = F64 ? (ST)P.abs_pri_tol64 : (ST)P.abs_pri_tol, dtol = F64 ? (ST)P.abs_dua_tol64 : (ST)P.abs_dua_tol;
    double dua_x = 0.0;
    ST pri_u = 0, dua_u = 0, pri_xf = 0, dua_xf = 0;                             // (XB: the state residuals in the slack's own type)
    // XB: knot k's state slack / dual from the rollout's x_k  (admm.cpp:46, 55-58, 68; q~ for :79-80)
    auto state_sets = [&](auto res_tag, auto kk, const double (&x)[NX]) {
        constexpr bool RES = decltype(res_tag)::value;
        constexpr int k = decltype(kk)::value;
#pragma unroll
        for (int m = 0; m < NX; ++m) {
            const ST xf = (ST)x[m];
            const ST t = xf + G[k][m];                                          // vnew = x + g
            const ST vn = clamp3(t, (ST)s_xb[k * 2 * NX + m], (ST)s_xb[k * 2 * NX + NX + m]);
            const ST gn = t - vn;                                               // g = g + x - vnew
            if constexpr (RES) {
                pri_xf = lean_max(pri_xf, lean_abs(xf - vn));
                dua_xf = lean_max(dua_xf, lean_abs((QT[k][m] + G[k][m]) - vn)); // v = the previous vnew = q~ + g
            }
            G[k][m] = gn;
            QT[k][m] = vn - gn;
        }
    };

    // ================= fused forward sweep: forward_pass (admm.cpp:25-35) + update_slack (:43-59) + update_dual (:65-69)
    // (+ RES: the residual maxima of termination_condition, :93-96) =================
    auto forward = [&](auto res_tag, bool first_iter) {
        constexpr bool RES = decltype(res_tag)::value;
        if constexpr (RES) {
            dua_x = 0.0, pri_u = 0, dua_u = 0, pri_xf = 0, dua_xf = 0;
            if constexpr (!XB)
                if (first_iter) {   // cold start: the previous state slack is the zero workspace at knot 0 too, where vnew is x0 (admm.cpp:94)
#pragma unroll
                    for (int m = 0; m < NX; ++m) dua_x = fmax(dua_x, fabs(X[0][m]));
                }
        }
        double xr[NX];                                                          // XB: the running x_k
#pragma unroll
        for (int m = 0; m < NX; ++m) xr[m] = X[0][m];
        sfor<0, N - 1>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            constexpr int kx = XB ? 0 : k;                                      // where x_k lives: the running vector, or the trajectory
            if constexpr (!UBK || XB) asm volatile("" ::: "memory");   // per-knot bounds are re-read from LDS at their knot, not hoisted out of the solve
            if constexpr (XB) state_sets(res_tag, kk, xr);
            const double (&xk)[NX] = XB ? xr : X[kx];
            double dk[NU], u[NU], xn[NX];
#pragma unroll
            for (int a = 0; a < NU; ++a) dk[a] = (double)D[k][a];
            // x+ = (A - B Kinf) x - B d: NX independent chains, none waits for u
#pragma unroll
            for (int m = 0; m < NX; ++m) {
                double acc = -(cB[m * NU] * dk[0]);
#pragma unroll
                for (int a = 1; a < NU; ++a) acc = fma(-cB[m * NU + a], dk[a], acc);
                xn[m] = acc;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int m = 0; m < NX; ++m) xn[m] = fma(cM[m * NX + j], xk[j], xn[m]);
            // u = -Kinf x - d
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                if constexpr (TMPC_LEAN_SPLITK && NX >= 4) {
                    double u0 = -dk[a], u1 = -(cK[a * NX + NX / 2] * xk[NX / 2]);
#pragma unroll
                    for (int j = 0; j < NX / 2; ++j) u0 = fma(-cK[a * NX + j], xk[j], u0);
#pragma unroll
                    for (int j = NX / 2 + 1; j < NX; ++j) u1 = fma(-cK[a * NX + j], xk[j], u1);
                    u[a] = u0 + u1;
                } else {
                    double acc = -dk[a];
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc = fma(-cK[a * NX + j], xk[j], acc);
                    u[a] = acc;
                }
            }
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                const ST uf = (ST)u[a];
                const ST t = uf + Y[k][a];                                      // znew = u + y  (admm.cpp:45)
                ST l_ = lo[a], h_ = hi[a];
                if constexpr (!UBK) l_ = (ST)s_bnd[k * 2 * NU + a], h_ = (ST)s_bnd[k * 2 * NU + NU + a];
                const ST zn = clamp3(t, l_, h_);                                //   clamped to [u_min, u_max]  (:50-52)
                Y[k][a] = t - zn;                                               // y = y + u - znew  (:67)
                if constexpr (RES) {
                    pri_u = lean_max(pri_u, lean_abs(uf - zn));                 // (:95)
                    dua_u = lean_max(dua_u, lean_abs(Z[k][a] - zn));            // (:96), times rho at the check
                }
                Z[k][a] = zn;
            }
#pragma unroll
            for (int m = 0; m < NX; ++m) {
                if constexpr (XB) {
                    xr[m] = xn[m];
                } else {
                    if constexpr (RES) dua_x = fmax(dua_x, fabs(X[k + 1][m] - xn[m]));   // v - vnew with v = the previous x  (:94)
                    X[k + 1][m] = xn[m];
                }
            }
            // (the residual maxima are only read under `!conv`: left alone, the compiler sinks the whole chain into that
            // branch, behind the sweep, and keeps every knot's u, znew and previous x alive for it — 190 spilled registers)
            if constexpr (RES) asm volatile("" : "+v"(pri_u), "+v"(dua_u), "+v"(dua_x), "+v"(pri_xf), "+v"(dua_xf));
            if constexpr (TMPC_LEAN_KNOT_BARRIER) __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (XB) {
            asm volatile("" ::: "memory");
            state_sets(res_tag, std::integral_constant<int, N - 1>{}, xr);      // the terminal knot's slack / dual
            if constexpr (RES) asm volatile("" : "+v"(pri_xf), "+v"(dua_xf));
        }
    };

    // ================= fused backward sweep: update_linear_cost (admm.cpp:75-83) + backward_pass_grad (:13-20), scaled by
    // -1 / rho; q, r, p never stored =================
    auto backward = [&]() {
        double p[NX];
        if constexpr (REFS == REF_SHARED) asm volatile("" ::: "memory");        // (reference terms: read from LDS where used)
#pragma unroll
        for (int m = 0; m < NX; ++m) {                                          // p~_{N-1} = vnew_{N-1} - g_{N-1} (+ Pinf' xref / rho)  (:81-82)
            p[m] = XB ? (double)QT[N - 1][m] : X[XB ? 0 : N - 1][m];
            if constexpr (REFS == REF_SHARED) p[m] += s_cpt[m];
        }
        sfor<0, N - 1>([&](auto kk) {
            constexpr int k = N - 2 - decltype(kk)::value;
            double r[NU], t[NU];
            if constexpr (REFS == REF_SHARED) asm volatile("" ::: "memory");
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                r[a] = (double)(Z[k][a] - Y[k][a]);                             // r~ = znew - y  (:77-78)
                if constexpr (REFS == REF_SHARED) r[a] += s_cr[k * NU + a];     //      + R~ uref / rho
                t[a] = r[a];
            }
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int a = 0; a < NU; ++a) t[a] = fma(cB[j * NU + a], p[j], t[a]);   // B' p~_{k+1} + r~_k
#pragma unroll
            for (int a = 0; a < NU; ++a) {                                      // d_k = Quu_inv (B' p_{k+1} + r_k)  (:17)
                double acc = cC[a * NU] * t[0];
#pragma unroll
                for (int c = 1; c < NU; ++c) acc = fma(cC[a * NU + c], t[c], acc);
                D[k][a] = (DT)acc;
            }
            if constexpr (k > 0) {                                              // (p_0 is never read)
                double ap[NX];
#pragma unroll
                for (int m = 0; m < NX; ++m) {                                  // q~_k - Kinf' r~_k
                    double acc = XB ? (double)QT[k][m] : X[XB ? 0 : k][m];     // q~_k = vnew_k - g_k  (:79-80)
                    if constexpr (REFS == REF_SHARED) acc += s_cq[k * NX + m];  //      + Q~ xref / rho
#pragma unroll
                    for (int a = 0; a < NU; ++a) acc = fma(-cK[a * NX + m], r[a], acc);
                    ap[m] = acc;
                }
#pragma unroll
                for (int j = 0; j < NX; ++j)
#pragma unroll
                    for (int m = 0; m < NX; ++m) ap[m] = fma(cM[j * NX + m], p[j], ap[m]);   // + AmBKt p~_{k+1}  (:18)
#pragma unroll
                for (int m = 0; m < NX; ++m) p[m] = ap[m];
            }
            if constexpr (TMPC_LEAN_KNOT_BARRIER) __builtin_amdgcn_sched_barrier(0);
        });
    };

    // solution = projected slack of the iteration (admm.cpp:187-188, :204-205); status of the instance.  This direct form
    // (every lane its own instance: scattered 16-byte pieces) serves the instances that converge inside the loop, a few at
    // a time; the final store below goes through LDS
    // element (k, m) of the solution: the state slack vnew — x itself, or q~ + g brought back inside the bounds it was clamped
    // to (fp32 rounding of the sum can leave them by an ulp)
    auto vnew_at = [&](auto kk, auto mm) -> float {
        constexpr int k = decltype(kk)::value, m = decltype(mm)::value;
        if constexpr (XB) return (float)clamp3(QT[k][m] + G[k][m], (ST)s_xb[k * 2 * NX + m], (ST)s_xb[k * 2 * NX + NX + m]);
        else return (float)X[XB ? 0 : k][m];
    };
    auto store = [&](bool solved_flag) {
        // one opaque base address per array, constant offsets behind it: left to itself the compiler forms the 99 store
        // addresses once, outside the iteration loop (the LIVE variant stores inside it), and spills 200 registers for them
        float *xo = P.xout + b * EX, *uo = P.uout + b * EU, *ro = P.res + b * 4;
        asm volatile("" : "+v"(xo), "+v"(uo), "+v"(ro));
        sfor<0, N>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            sfor<0, NX>([&](auto mm) { xo[k * NX + decltype(mm)::value] = vnew_at(kk, mm); });
            __builtin_amdgcn_sched_barrier(0);   // (a knot's conversions next to its stores, not eighty temporaries up front)
        });
#pragma unroll
        for (int k = 0; k < N - 1; ++k)
#pragma unroll
            for (int a = 0; a < NU; ++a) uo[k * NU + a] = (float)Z[k][a];
        P.iter[b] = P.iter_offset + it;
        P.solved[b] = solved_flag ? 1 : 0;
        ro[0] = (float)res0;
        ro[1] = (float)res1;
        ro[2] = (float)res2;
        ro[3] = (float)res3;
    };

    // Iterations whose termination check can matter carry the residual arithmetic (every check when the tolerances are
    // positive; otherwise nobody can converge and only the last check's values are ever reported); all others run in a
    // tight loop of their own, so that the two forms of the forward sweep never meet at a join (a join costs a copy per
    // loop-carried register and lets the compiler hoist their common parts above the branch, live across everything).
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int max_iter = P.max_iter;
    int i = 0;
    while (i < max_iter) {
        int next_res = max_iter;                                                // 0-based index of the next iteration with residuals
        if (ct > 0) {
            if (LIVE && can_converge) next_res = (i / ct) * ct + ct - 1;
            else if (last_check_it - 1 >= i) next_res = last_check_it - 1;
        }
        const int n_plain = (next_res < max_iter ? next_res : max_iter) - i;
        for (int j = 0; j < n_plain; ++j) {
            forward(std::false_type{}, false);
            backward();
        }
        i += n_plain;
        if (!LIVE || !conv) it += n_plain;                                      // admm.cpp:143
        if (i >= max_iter) break;
        forward(std::true_type{}, i == 0);
        i += 1;
        if (!LIVE || !conv) {                                                   // termination_condition (admm.cpp:89-107)
            it += 1;
            res0 = XB ? pri_xf : (ST)0;                                         // (no active state bound: x - vnew = 0)
            res1 = (XB ? dua_xf : (ST)dua_x) * rho;
            res2 = pri_u;
            res3 = dua_u * rho;
        }
        if constexpr (LIVE) {
            const bool now = active && !conv && res0 < ptol && res2 < ptol && res1 < dtol && res3 < dtol;
            if (now) {                                                          // returns before v = vnew and the backward pass (:181-193)
                store(true);
                conv = 1;
            }
            if (!__builtin_amdgcn_ballot_w64(active && !conv)) break;           // every instance of this wavefront finished
        }
        backward();
    }
    // ---- final store of every instance that has not stored at its convergence: through LDS, so that a store instruction
    // writes whole 64-byte pieces (X: 16 consecutive floats of 4 instances) or one contiguous 256 bytes (U) instead of 64
    // scattered 16-byte / 4-byte ones — with every wavefront finishing at once the scattered form took 50 us of a 300 us
    // launch (26 MB at 0.5 TB/s) ----
    {
        const bool mine = active && !conv;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(mine);
        if (mask) {
            const long w0 = (long)blockIdx.x * 256 + (tid & ~63);          // the wavefront's first instance
            store_wave_coalesced<EX, EU>(s_stage[tid >> 6], P.xout + w0 * EX, P.uout + w0 * EU, lane, mask,
                                         [&](auto ee) { constexpr int e = decltype(ee)::value; return vnew_at(std::integral_constant<int, e / NX>{}, std::integral_constant<int, e % NX>{}); },
                                         [&](auto ee) { constexpr int e = decltype(ee)::value; return (float)Z[e / NU][e % NU]; });
            if (mine) {
                float *ro = P.res + b * 4;
                P.iter[b] = P.iter_offset + it;
                P.solved[b] = 0;
                ro[0] = (float)res0, ro[1] = (float)res1, ro[2] = (float)res2, ro[3] = (float)res3;
#ifdef TMPC_LEAN_CLOCK_PROBE
                ro[0] = (float)(__builtin_amdgcn_s_memtime() - probe_t0);          // core clocks of the iteration loop (+ store issue)
                ro[1] = (float)(__builtin_amdgcn_s_memrealtime() - probe_r0);      // ... in 100 MHz ticks
                ro[2] = (float)(probe_entry & 0xFFFFFFull);                        // kernel entry on the chip-wide 100 MHz counter
#endif
            }
        }
    }

    {   // global status block: wavefront max of the residuals, count of unsolved instances
        float m0 = active ? (float)res0 : 0.f, m1 = active ? (float)res1 : 0.f, m2 = active ? (float)res2 : 0.f, m3 = active ? (float)res3 : 0.f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            m0 = fmaxf(m0, __shfl_xor(m0, o, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, o, 64));
            m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, o, 64));
        }
        const unsigned long long unsolved = __builtin_amdgcn_ballot_w64(active && !conv);
#ifdef TMPC_LEAN_CLOCK_PROBE
        if (active) P.iter[b] = (int)(__builtin_amdgcn_s_memrealtime() & 0xFFFFFFull);     // stores issued
        __builtin_amdgcn_s_waitcnt(0);
        if (active) P.solved[b] = (int)(__builtin_amdgcn_s_memrealtime() & 0xFFFFFFull);   // ... and acknowledged
#endif
        fold_status(P, m0, m1, m2, m3, __popcll(unsolved), tid);   // (one set of atomics per workgroup)
#ifdef TMPC_LEAN_CLOCK_PROBE
        if (active) P.res[b * 4 + 3] = (float)(__builtin_amdgcn_s_memrealtime() & 0xFFFFFFull);   // after the status fold
#endif
    }
}

}  // namespace tmpc
