// Fused ADMM kernel with the recurrences on the fp64 matrix cores, the duals in REGISTERS and only the array the two
// sweeps exchange in LDS: "mfmar<nx,nu,N>" — compile-time horizon, box bounds + affine dynamics term + second-order
// cones (BASELINE config 4: the rocket, N = 50).
//
// Why (on top of admm_mfmac.hip.h, whose algebra, lane mapping and operand packs this kernel shares): with every
// per-knot array in LDS a tile of 16 instances takes 73.5 KB at N = 50, a CU holds two tiles, and its four SIMDs run
// one wavefront each — every one of them alone with its latencies (measured: ~1400 cycles per knot and ADMM iteration,
// of which the matrix cores work 384).  The register file is the large on-chip memory (512 KB per SIMD against 160 KB
// of LDS per CU), and the duals are private to the wavefront that updates them: with the horizon a compile-time
// constant the knot loops of the forward sweep unroll and the duals of knot k are plain registers.  LDS keeps only
//   A3[position][row][instance]: the hand-over cell of every row — x_{k+1} / u_k on the way from the rollout to the
//   wavefront that owns the row's sets, the fused (slack - dual) sum on the way back to the backward sweep, t_k =
//   B'p + r from the backward sweep to the next rollout
// = 9 floats per knot for the rocket, 28 KB per tile at N = 50: four tiles (12 wavefronts, three per SIMD) per CU.
//
// Three wavefronts per tile, as in the LDS kernel, with the sets shared out so that each holds about the same number
// of duals (the register allocation of a kernel is the maximum over its wavefronts):
//   wave 0: the matrix products of both sweeps + the box set of state slot 1 (rows 4 ..: one dual per knot);
//   wave 1: state slot 0 (rows 0 .. 3): box set and the state cone (two duals per knot);
//   wave 2: the input rows: box set and the input cone (two duals per knot).
// The state cone must lie in slot 0 (rows 0 .. 3) — the host checks; other layouts stay on the LDS kernel.
#pragma once
#include "admm_mfmac.hip.h"

namespace tmpc {

template <int NX, int NU, int N>
struct RegShape {
    using S = ConeShape<NX, NU>;
    static constexpr int PLEN = 16 * S::NROW;   // floats of one position: the A3 cell of every row x 16 instances
    static constexpr size_t lds_bytes(int nk) {
        return sizeof(float) * ((size_t)PLEN * (N - 1) + 64 + ((S::bounds_len(nk) + 1) & ~1) + (((size_t)S::NROW * N + 2) & ~(size_t)1) + 128 + 200) +
               sizeof(double) * 8;
    }
};

// PF: the previous slack is read from HBM two pairs of knots ahead of its use (solves whose termination check is live:
// every checking iteration reads it) — or at its use (fixed-iteration solves: read once per solve; the read-ahead's
// registers and moves cost them 2 %, measured side by side).
template <int NX, int NU, int N, int REFS, int CX, int CU, bool BV, bool PF>
__global__ __launch_bounds__(192, 3) void admm_mfmar_kernel(const AdmmParams P) {
    // (three wavefronts per SIMD also for short horizons, whose 2 N dual registers would allow more: at N = 10 four and six
    // per SIMD measured slower, 1.22 / 1.59 ms against 1.12)
    using S = ConeShape<NX, NU>;
    constexpr int XS = S::XS, NROW = S::NROW, PLEN = RegShape<NX, NU, N>::PLEN;
    constexpr bool EXT = CX + CU > 0;
    static_assert(N >= 3, "horizon");
    extern __shared__ __align__(16) unsigned char s_raw_r[];
    constexpr int nk = BV ? N : 1;   // BV: the bounds depend on the knot (per-knot pack in LDS), else one knot's worth in registers
    float *s_state = reinterpret_cast<float *>(s_raw_r);
    float *s_pad = s_state + (size_t)PLEN * (N - 1);          // 64 zeros: what lanes without a row read and write
    float *s_bnd = s_pad + 64;
    float *s_ref = s_bnd + ((S::bounds_len(nk) + 1) & ~1);    // [N][NROW] and one zero cell behind (even offset: fp64 cells follow)
    double *s_pterm = reinterpret_cast<double *>(s_ref + (((size_t)NROW * N + 2) & ~(size_t)1));
    float *s_ring = reinterpret_cast<float *>(s_pterm + 8);   // pri_u[64] dua_u[64] (wave 2 -> 0)
    float *s_xchg = s_ring + 128;                             // pri_x[64] dua_x[64] (wave 1 -> 0), conv[64], any_left, [196] step counter
    __shared__ uint4 s_cmask[4];  // [lane group]: x head bits, x axis bits, u head bit, u axis bit (bit = slot)

    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, g = l >> 4, j = l & 15;
    // Persistent workgroups: a workgroup takes 16-instance tiles off a global counter until none is left (the launch
    // has at most as many workgroups as fit on the chip at once).  Left to the hardware dispatcher, 2 048 equally long
    // tiles on 1 024 slots took 2.5 rounds: workgroups are handed to the XCDs in strict rotation, so a slot freed on
    // one XCD stays empty while another XCD is still full (measured: scripts/mfmar_timeline.py).
    const int n_tiles = (P.batch + 15) / 16;
    __shared__ int s_tile;
    int tile = 0;
    long slot_id = j;                                          // of the current tile (set at the top of the tile loop)
    bool active = false;
    long b = 0;
    const long EX = (long)NX * N, EU = (long)NU * (N - 1);
    // rows of this lane: slot 0 -> x_g, slot 1 -> x_{4+g}, slot 2 -> u_g
    const int row0 = g, row1 = 4 + g, row2 = g;
    const bool ok0 = row0 < NX, ok1 = row1 < NX, ok2 = row2 < NU;

    // ---- stage constants ----
    for (int i = tid; i < S::bounds_len(nk); i += 192) s_bnd[i] = P.bounds[i];
    if constexpr (REFS == REF_SHARED) {
        // -(Xref .* Q~), -(Uref .* R~) as update_linear_cost forms them (admm.cpp:77-80), per knot
        for (int i = tid; i < NROW * N + 1; i += 192) {
            const int k = i / NROW, r = i % NROW;
            float v = 0.f;
            if (i < NROW * N) {
                if (r < NX) v = -(P.xref[k * NX + r] * P.bounds[2 * NROW * nk + r]);
                else if (k < N - 1) v = -(P.uref[k * NU + (r - NX)] * P.bounds[2 * NROW * nk + r]);
            }
            s_ref[i] = v;
        }
    }
    const double *gc64 = reinterpret_cast<const double *>(P.coef);
    if constexpr (REFS == REF_SHARED) {
        // terminal cost: -(Xref_{N-1}' Pinf)' (admm.cpp:81-82); Pinf follows the lane fields, row-major [NX][NX]
        const double *Pinf = gc64 + S::NF * 64;
        if (tid < NX) {
            double acc = 0.0;
            for (int c = 0; c < NX; ++c) acc = fma(Pinf[c * NX + l], (double)P.xref[(N - 1) * NX + c], acc);  // (Pinf^T xref)[l]
            s_pterm[l] = -acc;
        }
    }
    if constexpr (EXT) {
        if (tid < 4) {
            const int gg = tid;
            unsigned hx = 0u, ax = 0u, hu = 0u, au = 0u;
            if constexpr (CX > 0)
                for (int sl = 0; sl < 2; ++sl) {
                    const int row = 4 * sl + gg;
                    if (row < NX && row >= P.Acx[0] && row < P.Acx[0] + P.qcx[0] - 1) hx |= 1u << sl;
                    if (row < NX && row == P.Acx[0] + P.qcx[0] - 1) ax |= 1u << sl;
                }
            if constexpr (CU > 0) {
                if (gg < NU && gg >= P.Acu[0] && gg < P.Acu[0] + P.qcu[0] - 1) hu = 1u;
                if (gg < NU && gg == P.Acu[0] + P.qcu[0] - 1) au = 1u;
            }
            s_cmask[tid] = make_uint4(hx, ax, hu, au);
        }
    }
    constexpr int PAD_LO = 2 * NROW * nk + NROW, PAD_HI = PAD_LO + 1;
    __syncthreads();
    if (tid == 0) {
        s_bnd[PAD_LO] = -__builtin_inff();
        s_bnd[PAD_HI] = __builtin_inff();
    }
    __syncthreads();

    auto cone_scale = [&](float a2, float axv, float mu, float rmu, float &sc, float &ax_new) {
        // The public solver's cone "projection" (restated in oracle/: a <= -mu t -> 0; a <= mu t -> s; else
        // 1/2 (1 + mu t / a) (w, a / mu)), from the head norm^2 and the axis value summed over the instance's lanes:
        // the factor for the head rows and the new axis value
        const float an = __builtin_amdgcn_sqrtf(a2), u0 = axv * mu;
        const bool zero = an <= -u0, keep = !zero && an <= u0;
        const float half = 0.5f * (1.f + u0 * __builtin_amdgcn_rcpf(an));
        sc = zero ? 0.f : (keep ? 1.f : half);
        ax_new = zero ? 0.f : (keep ? axv : half * (an * rmu));
    };
    const float rho = P.rho;
    // ---- bounds of this lane's rows: registers when they do not depend on the knot, else the LDS pack per knot ----
    const int bidx[3] = {ok0 ? row0 : -1, ok1 ? row1 : -1, ok2 ? NX + row2 : -1};
    float lo_c[3], hi_c[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        lo_c[sl] = bidx[sl] < 0 ? -__builtin_inff() : s_bnd[bidx[sl]];
        hi_c[sl] = bidx[sl] < 0 ? __builtin_inff() : s_bnd[NROW + bidx[sl]];
    }
    // (rows a lane does not own read the +-inf pads: one unconditional LDS read, no lane masks)
    const int bl[3] = {bidx[0] < 0 ? PAD_LO : bidx[0], bidx[1] < 0 ? PAD_LO : bidx[1], bidx[2] < 0 ? PAD_LO : bidx[2]};
    const int bh[3] = {bidx[0] < 0 ? PAD_HI : NROW + bidx[0], bidx[1] < 0 ? PAD_HI : NROW + bidx[1], bidx[2] < 0 ? PAD_HI : NROW + bidx[2]};
    const int bs3[3] = {bidx[0] < 0 ? 0 : 2 * NROW, bidx[1] < 0 ? 0 : 2 * NROW, bidx[2] < 0 ? 0 : 2 * NROW};
    int opq_bv = 0;                                            // (see opaque_zero below: keeps the per-knot reads inside the iteration)
    auto lo_of = [&](int k, int sl) -> float {
        if constexpr (BV) return s_bnd[k * bs3[sl] + bl[sl] + opq_bv];
        else return lo_c[sl];
    };
    auto hi_of = [&](int k, int sl) -> float {
        if constexpr (BV) return s_bnd[k * bs3[sl] + bh[sl] + opq_bv];
        else return hi_c[sl];
    };

    // ---- LDS addressing: the cell of (row, instance j) of a position sits at row * 16 + j = (row base of the slot) * 16 + l;
    // lanes that do not own the slot's row point at the pad (stride 0), which holds exact zeros for the whole solve ----
    typedef float __attribute__((address_space(3))) lds_f;
    typedef volatile float __attribute__((address_space(3))) lds_vf;
    typedef volatile int __attribute__((address_space(3))) lds_vi;
    lds_f *const sbase = (lds_f *)s_state;
    const int rbase[3] = {0, 64, NX * 16};                     // slot row base * 16 (slot 1: rows 4 + g)
    const bool okr[3] = {ok0, ok1, ok2};
    lds_f *a_ptr[3];
    int a_str[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        a_ptr[sl] = sbase + (okr[sl] ? rbase[sl] + l : PLEN * (N - 1) + l);
        a_str[sl] = okr[sl] ? PLEN : 0;
    }
    // the affine term rides in the products: K index 11 (slot 2 of lane group 3, a row no shape uses: nu <= 3 there, else
    // it is added on the VALU) carries the constant 1 and the operand columns f / APf, BPf
    constexpr bool ONE_COL = NU <= 3;
    const bool one_lane = ONE_COL && g == 3;
    // reference pack [knot][row]: rows the lane does not own read the zero cell behind it
    const int rf_off[3] = {ok0 ? row0 : NROW * N, ok1 ? row1 : NROW * N, ok2 ? NX + row2 : NROW * N};
    const int rf_str[3] = {ok0 ? NROW : 0, ok1 ? NROW : 0, ok2 ? NROW : 0};
    (void)rf_off, (void)rf_str;
    // HBM scratch of this tile: cone slack kept around a check
    float *scr = P.scratch + l;                                // (+ the tile's block, set in the tile loop)

    int it = 0, conv = 0;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    const int ct = P.check_termination;
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;

    // membership of this lane's row in its side's cone, as 0 / 1 weights (slot 0 on the state side), and mu, 1 / mu
    float hw = 0.f, aw = 0.f, mu = 1.f, rmu = 1.f;
    if constexpr (EXT) {
        const uint4 mk = s_cmask[g];
        if (wave == 1 && CX > 0) mu = P.cx[0], rmu = 1.f / P.cx[0], hw = (float)(mk.x & 1u), aw = (float)(mk.y & 1u);
        if (wave == 2 && CU > 0) mu = P.cu[0], rmu = 1.f / P.cu[0], hw = (float)(mk.z & 1u), aw = (float)(mk.w & 1u);
    }
    // box (+ cone) sets of one row at one knot (admm.cpp:43-69): new slack(s) vn (vc), dual update(s), and the sum over
    // the sets of slack - dual that the backward sweep turns into q / r
    auto row_sets = [&](auto has_cone, float xf, float lo, float hi, float &a1, float &a2, float &vn, float &vc) -> float {
        vn = __builtin_amdgcn_fmed3f(xf + a1, lo, hi);                           // admm.cpp:52-56
        a1 = (a1 + xf) - vn;                                                     // admm.cpp:68
        float s = vn - a1;
        if constexpr (decltype(has_cone)::value) {
            vc = xf + a2;
            const float n2 = mfc_inst_sum(hw * vc * vc), axv = mfc_inst_sum(aw * vc);
            float sc, ax_new;
            cone_scale(n2, axv, mu, rmu, sc, ax_new);
            vc = hw != 0.f ? vc * sc : vc;
            vc = aw != 0.f ? ax_new : vc;
            a2 = (a2 + xf) - vc;
            s += vc - a2;
        }
        return s;
    };

    double x0r[2] = {0.0, 0.0};

    lds_vi *const s_step = (lds_vi *)reinterpret_cast<int *>(s_xchg + 196);
    int any_left = 1;                                          // some instance of the tile still iterates (all waves agree)
#ifdef TMPC_MFMAC_PROBE
    // timing probe (scripts/mfmac_cycles.py; results are NOT a solution): TINYMPC_HIP_MFMAC_DEBUG & 8 reports s_memtime
    // deltas per knot step in place of the residuals
    const bool probe = (P.mpc_steps & 8) != 0;
    const bool probe_tl = (P.mpc_steps & 32) != 0;           // timeline: start (100 MHz clock, 2 x 16 bits), duration, where
    long long T_fwd = 0, T_bar = 0, T_bwd = 0, T_w1 = 0;
#define TMPC_PROBE(x) x
#else
#define TMPC_PROBE(x)
#endif
    // what an iteration has to do besides iterating (identical in the three wavefronts)
    struct Flags {
        bool need_res, write_old, last, write_sol, read_old, full;
        int step0;                                             // the step counter's value before this iteration's sweep
    };
    auto flags_of = [&](int i) -> Flags {
        Flags F;
        const int itn = i + 1;
        const bool check = ct > 0 && itn % ct == 0;
        F.need_res = check && (can_converge || itn == last_check_it);
        const bool check_next = ct > 0 && (itn + 1) % ct == 0 && itn < P.max_iter;
        F.write_old = check_next && (can_converge || itn + 1 == last_check_it);   // the next iteration reads this one's slack
        F.last = itn == P.max_iter;
        F.write_sol = F.write_old || F.need_res || F.last;     // box slack -> xout / uout
        F.read_old = F.need_res && itn > 1;                    // the zero workspace before the first iteration
        F.full = F.need_res || F.write_sol;
        F.step0 = i * (N - 1);
        return F;
    };
    // residual terms and solution / scratch traffic of one row at one knot, on the iterations around a check
    // (admm.cpp:93-96: pri = |x - vnew|, dua = |v - vnew| rho over every set)
    // The previous slack of a row at a knot comes back from HBM (solution buffers / scratch) on a checking iteration: it is
    // loaded one pair of knots AHEAD of its use (load_old), so that the sets wavefronts do not sit out the memory latency
    // at every knot (with the check live every iteration that was half of the solve time).
    struct Old {
        float box, cone;
    };
    auto load_old = [&](auto has_cone, const Flags &F, const float *out_cell, bool own, const float *scr_cell) -> Old {
        Old o = {0.f, 0.f};
        if (F.read_old) {
            if (active && own) o.box = *out_cell;
            if constexpr (decltype(has_cone)::value) o.cone = *scr_cell;
        }
        return o;
    };
    auto around_check = [&](auto has_cone, const Flags &F, float &pri, float &dua, float xf, float vn, float vc, float *out_cell,
                            bool own, float *scr_cell, const Old &old) {
        if (F.need_res) {
            pri = fmaxf(pri, fabsf(xf - vn));
            dua = fmaxf(dua, fabsf(old.box - vn));
            if constexpr (decltype(has_cone)::value) {
                pri = fmaxf(pri, fabsf(xf - vc));
                dua = fmaxf(dua, fabsf(old.cone - vc));
            }
        }
        if (F.write_sol && active && !conv && own) *out_cell = vn;       // a converged instance's outputs are frozen
        if constexpr (decltype(has_cone)::value) {
            if (F.write_old) *scr_cell = vc;
        }
    };
    constexpr std::integral_constant<bool, (CX > 0)> cone_x{};
    constexpr std::integral_constant<bool, (CU > 0)> cone_u{};
    constexpr std::false_type no_cone{};
    // (an opaque zero, new every iteration: with it in the base addresses the per-knot addresses of the unrolled sweeps
    // are not loop invariants the compiler would hoist into ~150 registers and spill)
    auto opaque_zero = [&]() -> int {
        int z = 0;
        asm volatile("" : "+s"(z));
        return z;
    };

    float fm0 = 0.f, fm1 = 0.f, fm2 = 0.f, fm3 = 0.f;          // over this workgroup's tiles: residual maxima, unsolved instances
    int f_unsolved = 0;
    for (;;) {
    if (tid == 0) s_tile = (int)atomicAdd(&P.gacc[6], 1u);
    __syncthreads();
    tile = s_tile;
    if (tile >= n_tiles) break;
    slot_id = (long)tile * 16 + j;
    active = slot_id < P.batch;
    b = active ? slot_id : 0;
    scr = P.scratch + (size_t)tile * S::scratch_floats(N) + l;
    x0r[0] = (active && ok0) ? (double)P.x0[b * NX + row0] : 0.0;
    x0r[1] = (active && ok1) ? (double)P.x0[b * NX + row1] : 0.0;
    it = 0, conv = 0, any_left = 1;
    res0 = res1 = res2 = res3 = 0.f;
    // cold start = the zero workspace tiny_setup leaves (tiny_api.cpp:73-88)
    for (int i = tid; i < PLEN * (N - 1) + 64; i += 192) s_state[i] = 0.f;
    if (tid == 0) *s_step = 0;                                 // the hand-over step counter
    TMPC_PROBE(const unsigned long long tl0 = wall_clock64();)
    __syncthreads();
    if (wave != 0) {
        // ================= wave 1: state slot 0, wave 2: the input rows — box + cone of one row per lane =================
        // (admm.cpp:43-69, :93-96).  The duals of knot k are registers a1[k], a2[k] ([N-1]: state knot 0, which x0 feeds).
        float a1[N], a2[N];
#pragma unroll
        for (int k = 0; k < N; ++k) a1[k] = 0.f, a2[k] = 0.f;
        const bool st_side = wave == 1;
        const int sl = st_side ? 0 : 2;
        const bool own = st_side ? ok0 : ok2;
        int seen = 0;                                          // the step counter as this wavefront last read it
        for (int i = 0; i < P.max_iter; ++i) {
            const Flags F = flags_of(i);
            const int opq = opaque_zero();
            opq_bv = opq;
            float pri = 0.f, dua = 0.f, vn, vc = 0.f;
            float *const scr_i = scr + opq + sl * 64;
            lds_f *pa = (st_side ? a_ptr[0] : a_ptr[2]) + opq;
            if (st_side) {
                float *const xo_i = P.xout + b * EX + row0 + opq;
                {
                    const float xf = (float)x0r[0];                              // knot 0: its fused value feeds nothing
                    Old o0 = {0.f, 0.f};
                    if (F.full) o0 = load_old(cone_x, F, xo_i, own, scr_i + (N - 1) * 192);
                    (void)row_sets(cone_x, xf, lo_of(0, 0), hi_of(0, 0), a1[N - 1], a2[N - 1], vn, vc);
                    if (F.full) around_check(cone_x, F, pri, dua, xf, vn, vc, xo_i, own, scr_i + (N - 1) * 192, o0);
                }
                Old oA = {0.f, 0.f}, oB = {0.f, 0.f}, pA = oA, pB = oA;          // previous slack of this pair's knots, of the next pair's
                if (PF && F.full) {
                    oA = load_old(cone_x, F, xo_i + 1 * NX, own, scr_i);
                    oB = load_old(cone_x, F, xo_i + 2 * NX, own, scr_i + 192);
                    if (N - 1 > 2) pA = load_old(cone_x, F, xo_i + 3 * NX, own, scr_i + 2 * 192);
                    if (N - 1 > 3) pB = load_old(cone_x, F, xo_i + 4 * NX, own, scr_i + 3 * 192);
                }
                // Knots are taken TWO at a time: a knot's set arithmetic is one dependent chain of ~50 VALU instructions,
                // and a wavefront issues dependent instructions every 8-12 cycles but independent ones every ~6
                // (experiments/valu_dep_probe.hip) — two knots interleave into one stream.  Wave 0 is usually steps
                // ahead; then the next pair's cells are read now, under this pair's arithmetic.
                bool have = false;
                float xa0 = 0.f, xa1 = 0.f;                                      // cells read ahead
                auto until_step = [&](int want) {
                    TMPC_PROBE(const long long tb0 = probe ? clock64() : 0;)
                    while (seen < want) {
                        seen = __builtin_amdgcn_readfirstlane(*s_step);
                        if (seen < want) __builtin_amdgcn_s_sleep(1);
                    }
                    TMPC_PROBE(if (probe) T_w1 += clock64() - tb0;)
                };
#pragma unroll
                for (int k = 0; k < N - 1; k += 2) {
                    const bool two = k + 1 < N - 1;                              // (the last knot of an odd count goes alone)
                    const int want = F.step0 + k + (two ? 2 : 1);                // x_{k+1}, x_{k+2} are in the cells of positions k, k + 1
                    Old nA = {0.f, 0.f}, nB = {0.f, 0.f};                        // ... of the pair after next: read now (two pairs of latency cover)
                    if (PF && F.full) {
                        if (k + 4 < N - 1) nA = load_old(cone_x, F, xo_i + (k + 5) * NX, own, scr_i + (k + 4) * 192);
                        if (k + 5 < N - 1) nB = load_old(cone_x, F, xo_i + (k + 6) * NX, own, scr_i + (k + 5) * 192);
                    }
                    float xf0 = xa0, xf1 = xa1;
                    if (!have) {
                        until_step(want);
                        xf0 = *(lds_vf *)pa;
                        if (two) xf1 = *(lds_vf *)(pa + a_str[0]);
                    }
                    const int nxt = k + 2 < N - 1 ? (k + 3 < N - 1 ? 2 : 1) : 0;  // knots of the next pair
                    have = nxt > 0 && seen >= want + nxt;
                    if (have) {
                        xa0 = *(lds_vf *)(pa + 2 * a_str[0]);
                        if (nxt == 2) xa1 = *(lds_vf *)(pa + 3 * a_str[0]);
                    }
                    float vn0, vc0 = 0.f, vn1 = 0.f, vc1 = 0.f, s1 = 0.f;
                    const float s0 = row_sets(cone_x, xf0, lo_of(k + 1, 0), hi_of(k + 1, 0), a1[k], a2[k], vn0, vc0);
                    if (two) s1 = row_sets(cone_x, xf1, lo_of(k + 2, 0), hi_of(k + 2, 0), a1[k + 1], a2[k + 1], vn1, vc1);
                    if (F.full) {
                        if constexpr (!PF) {
                            oA = load_old(cone_x, F, xo_i + (k + 1) * NX, own, scr_i + k * 192);
                            if (two) oB = load_old(cone_x, F, xo_i + (k + 2) * NX, own, scr_i + (k + 1) * 192);
                        }
                        around_check(cone_x, F, pri, dua, xf0, vn0, vc0, xo_i + (k + 1) * NX, own, scr_i + k * 192, oA);
                        if (two) around_check(cone_x, F, pri, dua, xf1, vn1, vc1, xo_i + (k + 2) * NX, own, scr_i + (k + 1) * 192, oB);
                    }
                    if constexpr (PF) oA = pA, oB = pB, pA = nA, pB = nB;
                    *pa = s0;
                    if (two) pa[a_str[0]] = s1;
                    pa += 2 * a_str[0];
                }
            } else {
                float *const uo_i = P.uout + b * EU + row2 + opq;
                bool have = false;
                float ua0 = 0.f, ua1 = 0.f;
                Old oA = {0.f, 0.f}, oB = {0.f, 0.f}, pA = oA, pB = oA;
                if (PF && F.full) {
                    oA = load_old(cone_u, F, uo_i, own, scr_i);
                    oB = load_old(cone_u, F, uo_i + NU, own, scr_i + 192);
                    if (N - 1 > 2) pA = load_old(cone_u, F, uo_i + 2 * NU, own, scr_i + 2 * 192);
                    if (N - 1 > 3) pB = load_old(cone_u, F, uo_i + 3 * NU, own, scr_i + 3 * 192);
                }
#pragma unroll
                for (int k = 0; k < N - 1; k += 2) {
                    const bool two = k + 1 < N - 1;
                    const int want = F.step0 + k + (two ? 2 : 1);                // u_k, u_{k+1} are in the cells of positions k, k + 1
                    Old nA = {0.f, 0.f}, nB = {0.f, 0.f};
                    if (PF && F.full) {
                        if (k + 4 < N - 1) nA = load_old(cone_u, F, uo_i + (k + 4) * NU, own, scr_i + (k + 4) * 192);
                        if (k + 5 < N - 1) nB = load_old(cone_u, F, uo_i + (k + 5) * NU, own, scr_i + (k + 5) * 192);
                    }
                    float uf0 = ua0, uf1 = ua1;
                    if (!have) {
                        while (seen < want) {
                            seen = __builtin_amdgcn_readfirstlane(*s_step);
                            if (seen < want) __builtin_amdgcn_s_sleep(1);
                        }
                        uf0 = *(lds_vf *)pa;
                        if (two) uf1 = *(lds_vf *)(pa + a_str[2]);
                    }
                    const int nxt = k + 2 < N - 1 ? (k + 3 < N - 1 ? 2 : 1) : 0;
                    have = nxt > 0 && seen >= want + nxt;
                    if (have) {
                        ua0 = *(lds_vf *)(pa + 2 * a_str[2]);
                        if (nxt == 2) ua1 = *(lds_vf *)(pa + 3 * a_str[2]);
                    }
                    float vn0, vc0 = 0.f, vn1 = 0.f, vc1 = 0.f, s1 = 0.f;
                    const float s0 = row_sets(cone_u, uf0, lo_of(k, 2), hi_of(k, 2), a1[k], a2[k], vn0, vc0);
                    if (two) s1 = row_sets(cone_u, uf1, lo_of(k + 1, 2), hi_of(k + 1, 2), a1[k + 1], a2[k + 1], vn1, vc1);
                    if (F.full) {
                        if constexpr (!PF) {
                            oA = load_old(cone_u, F, uo_i + k * NU, own, scr_i + k * 192);
                            if (two) oB = load_old(cone_u, F, uo_i + (k + 1) * NU, own, scr_i + (k + 1) * 192);
                        }
                        around_check(cone_u, F, pri, dua, uf0, vn0, vc0, uo_i + k * NU, own, scr_i + k * 192, oA);
                        if (two) around_check(cone_u, F, pri, dua, uf1, vn1, vc1, uo_i + (k + 1) * NU, own, scr_i + (k + 1) * 192, oB);
                    }
                    if constexpr (PF) oA = pA, oB = pB, pA = nA, pB = nB;
                    *pa = s0;
                    if (two) pa[a_str[2]] = s1;
                    pa += 2 * a_str[2];
                }
            }
            if (F.need_res) {                                                    // hand the maxima to wave 0
                float *const dst = st_side ? s_xchg : s_ring;
                dst[l] = mf_inst_max(pri);
                dst[64 + l] = mf_inst_max(dua);
            }
            __syncthreads();                                                     // end of the forward sweep
            if (F.need_res) {
                __syncthreads();                                                 // wave 0 has decided
                conv = (int)s_xchg[128 + l];
                any_left = (int)s_xchg[192];
            }
            it += 1;
            if (F.last || !any_left) break;                                      // (the backward sweep is wave 0's)
        }
    } else {
        // ================= wave 0: the rollout (admm.cpp:25-35), the box set of state slot 1, the backward sweep =================
        // x+ = (A - B Kinf) x - B Quu_inv t + f,  u = -Kinf x - Quu_inv t  with t = B'p + r kept by the backward sweep
        // (d = Quu_inv t of admm.cpp:17 is never formed on its own: Quu_inv rides in the forward operand, f in the
        // column of the constant 1).  The product with t does not depend on x: the one of step k + 1 is issued behind the
        // x products of step k; the slot 1 work of step k - 1 runs underneath them.
        double cf[S::NF];
#pragma unroll
        for (int f = 0; f < S::NF; ++f) cf[f] = gc64[f * 64 + l];
        float a1[N];                                           // duals of the box set of slot 1 ([N-1]: knot 0)
#pragma unroll
        for (int k = 0; k < N; ++k) a1[k] = 0.f;
        for (int i = 0; i < P.max_iter; ++i) {
            const Flags F = flags_of(i);
            const int opq = opaque_zero();
            opq_bv = opq;
            float pri = 0.f, dua = 0.f;
            {
                TMPC_PROBE(const long long tf0 = probe ? clock64() : 0;)
                float *const xo_i = P.xout + b * EX + row1 + opq;
                auto t_product = [&](float t) -> mf_d4 {
                    mf_d4 c = {0.0, 0.0, 0.0, 0.0};
                    if constexpr (!ONE_COL) c[0] = cf[S::F_FD0], c[1] = cf[S::F_FD1];
                    return mf_mma(cf[S::F_MF2], (double)(one_lane ? 1.f : t), c);
                };
                float *const scr_i = scr + opq + 64;
                auto slot1_old = [&](int kn) -> Old {                            // the previous slack of knot kn's slot 1 rows (read a step ahead)
                    Old o = {0.f, 0.f};
                    if (F.full && F.read_old) {
                        if (active && ok1) o.box = xo_i[kn * NX];
                        if constexpr (CX > 0) o.cone = scr_i[(kn > 0 ? kn - 1 : N - 1) * 192];
                    }
                    return o;
                };
                auto slot1 = [&](int kn, float xf, float &dual, lds_f *cell, const Old &old) {   // knot kn of state slot 1; cell: where s goes
                    float vn, vc = 0.f, none = 0.f;
                    float s = row_sets(no_cone, xf, lo_of(kn, 1), hi_of(kn, 1), dual, none, vn, vc);
                    if (F.full) around_check(no_cone, F, pri, dua, xf, vn, vc, xo_i + kn * NX, ok1, nullptr, old);
                    if constexpr (CX > 0) {
                        // With a state cone enabled EVERY state row carries the cone set's slack and dual (the solver's
                        // arrays are full size); for a row outside the cone the "projection" is the identity: slack =
                        // x + dual, the dual stays zero, and the set contributes x to the fused value and |x_old - x|
                        // to the dual residual.
                        s += xf;
                        if (F.full) {
                            float *const sc = scr_i + (kn > 0 ? kn - 1 : N - 1) * 192;
                            if (F.need_res) dua = fmaxf(dua, fabsf(old.cone - xf));
                            if (F.write_old) *sc = xf;
                        }
                    }
                    if (cell) *cell = s;
                };
                lds_f *ph[3] = {a_ptr[0] + opq, a_ptr[1] + opq, a_ptr[2] + opq}; // cells of position k
                mf_d4 cpre = t_product(*ph[2]);
                float t_next = ph[2][a_str[2]];                                  // t of position 1
                double xa = x0r[0], xb = x0r[1];
                Old s1o = slot1_old(0);
                if constexpr (XS == 2) {
                    slot1(0, (float)x0r[1], a1[N - 1], nullptr, s1o);
                    if constexpr (PF) s1o = slot1_old(1);
                }
                float x1_prev = 0.f;
#pragma unroll
                for (int k = 0; k < N - 1; ++k) {
                    mf_d4 c = mf_mma(cf[S::F_MF0], xa, cpre);
                    if constexpr (XS == 2) c = mf_mma(cf[S::F_MF1], xb, c);
                    if (k + 1 < N - 1) {
                        cpre = t_product(t_next);
                        if (k + 2 < N - 1) t_next = ph[2][2 * a_str[2]];
                    }
                    if constexpr (XS == 2) {
                        if (k > 0) {
                            if constexpr (!PF) s1o = slot1_old(k);
                            slot1(k, x1_prev, a1[k - 1], ph[1] - a_str[1], s1o);
                            if constexpr (PF) s1o = slot1_old(k + 1);
                        }
                    }
                    xa = c[0], xb = c[1];
                    *(lds_vf *)ph[0] = (float)xa;                                // x_{k+1} (slot 0) for wave 1
                    *(lds_vf *)ph[2] = (float)c[2];                              // u_k for wave 2 (t_k is spent)
                    *s_step = F.step0 + k + 1;
                    x1_prev = (float)xb;
#pragma unroll
                    for (int q = 0; q < 3; ++q) ph[q] += a_str[q];
                }
                if constexpr (XS == 2) {
                    if constexpr (!PF) s1o = slot1_old(N - 1);
                    slot1(N - 1, x1_prev, a1[N - 2], ph[1] - a_str[1], s1o);
                }
                TMPC_PROBE(if (probe) T_fwd += clock64() - tf0;)
            }
            TMPC_PROBE(const long long tb0 = probe ? clock64() : 0;)
            __syncthreads();                                                     // end of the forward sweep: s of every knot is in LDS
            TMPC_PROBE(if (probe) T_bar += clock64() - tb0;)
            it += 1;
            if (F.need_res) {
                const float r0 = fmaxf(s_xchg[l], mf_inst_max(pri)), r1 = fmaxf(s_xchg[64 + l], mf_inst_max(dua)) * rho;
                const float r2 = s_ring[l], r3 = s_ring[64 + l] * rho;
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
                any_left = __builtin_amdgcn_ballot_w64(active && !conv) != 0ull;
                s_xchg[128 + l] = (float)conv;
                if (l == 0) s_xchg[192] = (float)any_left;
                __syncthreads();
            }
            if (F.last || !any_left) break;
            // ---------------- fused backward sweep (admm.cpp:75-83, :13-20) ----------------
            TMPC_PROBE(const long long tq0 = probe ? clock64() : 0;)
            lds_f *qa[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) qa[q] = a_ptr[q] + (N - 2) * a_str[q];
            double p[2], r_held;
            {
                double pt0 = 0.0, pt1 = 0.0;
                if constexpr (REFS == REF_SHARED) {
                    pt0 = ok0 ? s_pterm[row0] : 0.0;
                    pt1 = ok1 ? s_pterm[row1] : 0.0;
                }
                p[0] = pt0 - (double)(rho * *qa[0]);                             // admm.cpp:81-82
                p[1] = pt1 - (double)(rho * *qa[1]);
                float rr = 0.f;
                if constexpr (REFS == REF_SHARED) rr = s_ref[(N - 2) * rf_str[2] + rf_off[2]];
                r_held = (double)(rr - rho * *qa[2]);                            // admm.cpp:77-78
            }
            // A stage (position i2) produces p and t of knot i2 + 1:  p- = q + AmBKt p - Kinf' r (+ AmBKt Pinf f),
            // t = B'p + r (+ B' Pinf f), from q_{i2+1} (s of the state slots at position i2), r_{i2+1} (s of the input
            // slot one position up, held from the stage before) and the p of the stage before; the affine constants ride
            // in the column of the constant 1 (nu = 4: added here).  Only the products with p are on the recurrence: a
            // stage's accumulator start {q, r} and its product with r are formed and issued one stage ahead, behind the
            // previous stage's p products, from operands read from LDS two stages ahead.
            auto stage_operands = [&](int i2, float (&sv)[3], float (&rf)[3]) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    qa[q] -= a_str[q];
                    sv[q] = *qa[q];
                }
                if constexpr (REFS == REF_SHARED) {
                    rf[0] = s_ref[(i2 + 1) * rf_str[0] + rf_off[0]];
                    rf[1] = s_ref[(i2 + 1) * rf_str[1] + rf_off[1]];
                    rf[2] = s_ref[i2 * rf_str[2] + rf_off[2]];
                } else {
                    rf[0] = rf[1] = rf[2] = 0.f;
                }
            };
            auto stage_start = [&](int i2, const float (&sv)[3], const float (&rf)[3], double r_in) -> mf_d4 {
                mf_d4 c = {(double)(rf[0] - rho * sv[0]), (double)(rf[1] - rho * sv[1]), r_in, 0.0};
                if (i2 < 0) c[0] = c[1] = 0.0;                                   // q_0 enters p_0 only, which nothing reads
                if constexpr (!ONE_COL) c[0] += cf[S::F_APF0], c[1] += cf[S::F_APF1], c[2] += cf[S::F_BPF];
                return mf_mma(cf[S::F_MB2], one_lane ? 1.0 : r_in, c);           // [-Kinf^T; 0] r
            };
            float sv[3] = {0.f, 0.f, 0.f}, rf[3] = {0.f, 0.f, 0.f};
            lds_f *tw = a_ptr[2] + (N - 2) * a_str[2];                           // where t of the stage's knot goes
            stage_operands(N - 3, sv, rf);
            mf_d4 cpre = stage_start(N - 3, sv, rf, r_held);
            r_held = (double)(rf[2] - rho * sv[2]);                              // r of knot N - 3, for the stage after
            if (N >= 4) stage_operands(N - 4, sv, rf);
            for (int i2 = N - 3; i2 >= -1; --i2) {
                mf_d4 c = mf_mma(cf[S::F_MB0], p[0], cpre);                      // + [AmBKt; B^T] p
                if constexpr (XS == 2) c = mf_mma(cf[S::F_MB1], p[1], c);
                if (i2 >= 0) {
                    cpre = stage_start(i2 - 1, sv, rf, r_held);
                    r_held = (double)(rf[2] - rho * sv[2]);
                    if (i2 >= 2) stage_operands(i2 - 2, sv, rf);
                }
                p[0] = c[0], p[1] = c[1];
                *tw = (float)c[2];
                tw -= a_str[2];
            }
            TMPC_PROBE(if (probe) T_bwd += clock64() - tq0;)
        }
    }
#ifdef TMPC_MFMAC_PROBE
    if (probe_tl) {
        const unsigned long long tl1 = wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        res0 = (float)(tl0 & 0xFFFFull), res1 = (float)((tl0 >> 16) & 0xFFFFFFull), res2 = (float)(tl1 - tl0);
        res3 = (float)(((hw >> 8) & 15u) + 16u * ((hw >> 13) & 7u) + 128u * (xcc & 15u));   // cu_id | se_id | xcc_id
    }
    if (probe) {
        if (wave == 1) s_xchg[l] = (float)T_w1;
        __syncthreads();
        const float steps = (float)it * (float)(N - 1);
        res0 = (float)T_fwd / steps, res1 = (float)T_bar / steps, res2 = (float)T_bwd / steps, res3 = s_xchg[l] / steps;
    }
#endif
#undef TMPC_PROBE

    if (wave == 0 && active && !conv && g == 0) {
        P.iter[b] = P.iter_offset + it;
        P.solved[b] = 0;
    }
    if (wave == 0 && active && g == 0) {
        P.res[b * 4 + 0] = res0;
        P.res[b * 4 + 1] = res1;
        P.res[b * 4 + 2] = res2;
        P.res[b * 4 + 3] = res3;
    }
    {
        const bool rep = active && wave == 0;                // the other wavefronts report nothing
        fm0 = fmaxf(fm0, rep ? res0 : 0.f), fm1 = fmaxf(fm1, rep ? res1 : 0.f);
        fm2 = fmaxf(fm2, rep ? res2 : 0.f), fm3 = fmaxf(fm3, rep ? res3 : 0.f);
        f_unsolved += __popcll(__builtin_amdgcn_ballot_w64(rep && !conv && g == 0));
    }
    __syncthreads();                                           // the tile's LDS is free for the next one
    }   // tile loop
    {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            fm0 = fmaxf(fm0, __shfl_xor(fm0, off, 64));
            fm1 = fmaxf(fm1, __shfl_xor(fm1, off, 64));
            fm2 = fmaxf(fm2, __shfl_xor(fm2, off, 64));
            fm3 = fmaxf(fm3, __shfl_xor(fm3, off, 64));
        }
        fold_status(P, fm0, fm1, fm2, fm3, f_unsolved, tid);
    }
}

}  // namespace tmpc
