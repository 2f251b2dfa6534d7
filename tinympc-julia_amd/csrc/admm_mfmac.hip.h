// Fused ADMM kernel with the recurrences on the fp64 matrix cores, the per-instance state in LDS, rolled knot loops:
// "mfmac<nx,nu>" — run-time horizon, box bounds + affine dynamics term + second-order cones (BASELINE config 4).
//
// Why: the run-time-horizon stream kernel (admm_streamg.hip.h) keeps every trajectory in HBM and moves ~12 KB per
// instance and ADMM iteration through it (40 GB per launch for config 4: 340 x the algorithmic bytes).  Only the
// instances that are being iterated need their state on chip, and a one-shot solve needs little of it:
//   per knot and row   one dual per constraint set (g | y, gc | yc),
//                      ONE fused array handed from the forward to the backward sweep: sum over sets of (slack - dual),
//                      which for input rows shares its slot with the feed-forward term (t_k = B'p + r is written by the
//                      backward sweep exactly where su_k was read, and read by the forward sweep before su_k is written)
// = 24 floats per knot for the rocket (6 state rows, 3 of them in a cone, 3 input rows in a cone): 4.7 KB per instance
// at N = 50, so 16 instances — one wavefront — take 75 KB of LDS and a CU holds two wavefronts' worth.  Nothing of it
// ever goes to HBM: traffic is x0 in, the solution out, plus (see "residuals") the previous slack around a check.
//
// Mapping: one wavefront = 16 instances (the N side of a 16x16x4 fp64 MFMA tile), lane l = 16 g + j works on instance
// j.  The stacked vector [x; u] occupies tile rows / K indices  x_c -> c (c < 8),  u_a -> 8 + a, i.e. with the
// instruction's layout (A: lane l holds A[l % 16][l / 16]; B: B[l / 16][l % 16]; D: register v of lane l holds
// D[4 v + l / 16][l % 16], experiments/mfma_probe.hip) lane group g carries in its three "slots"
//     slot 0: x_g      slot 1: x_{4+g}      slot 2: u_g
// and a product's result registers are the next product's B operands as they stand: no cross-lane move on the chain.
//     forward : c = {f, 0};  c += M_t t;  c += M_x0 x[0];  c += M_x1 x[1]        -> c[0..1] = x+,  c[2] = u = -Kinf x - d
//               (M = [A - B Kinf, -B Quu_inv; -Kinf, -Quu_inv] applied to [x; t]: the feed-forward d = Quu_inv t of
//               admm.cpp:17 is never formed, t = B'p + r is what the backward sweep leaves; the product with t does not
//               wait for x)
//     backward: c = {q + APf, r + BPf};  c += N_u r;  c += N_x0 p[0];  c += N_x1 p[1]   -> c[0..1] = p-,  c[2] = t
//               (N = [AmBKt, -Kinf'; B', 0])
// The x slots of position k hold knot k + 1 (the rollout produces x_{k+1} together with u_k), so both sweeps index the
// state with ONE wave-uniform position; the backward sweep reads position i for q_{i+1} (used at once) and r_i (used one
// stage later).  Knot 0 of the state side (x0 is given) is handled once per iteration outside the loop.
//
// Cones may span lane groups (their rows are consecutive components): squared head norms and the axis value are
// summed over the four lanes of an instance (two xor-shuffles each), off the recurrence chain.
//
// Residuals (termination_condition, admm.cpp:89-107) need the PREVIOUS iteration's slack of every set, which this
// layout does not keep.  On an iteration that precedes a check the forward sweep therefore also writes the box slack to
// the solution buffers (xout / uout — where the slack of the last executed iteration has to end up anyway) and the cone
// slack to an HBM scratch, and the checking iteration reads them back: once per solve for the fixed-iteration benchmark
// configs, every check_termination-th iteration otherwise.
//
// Linear-inequality rows (LIN; `Alin_x x <= blin_x`, `Alin_u u <= blin_u` at every knot, bindings.cpp:414-450) are a third
// set per side with ONE slack and ONE dual array like the others: its dual takes NX (NU) more rows of a position, the
// projection is one instance sum (the row's dot product) per row, rows one after the other as the solver applies them.
//
// Scope: one-shot solves (cold start, workspace not kept), shared or zero references, at most 8 state / 4 input rows,
// box bounds, up to two cones per side and linear rows as the C-ABI takes them; no per-instance families, no adaptive rho
// (those stay on the stream / generic kernels).  Precision as everywhere: fp64 recurrences, fp32 state.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "admm_params.h"
#include "admm_mfma.hip.h"

namespace tmpc {

template <int NX, int NU>
struct ConeShape {
    static_assert(NX >= 1 && NX <= 8 && NU >= 1 && NU <= 4, "mfmac kernel: nx <= 8, nu <= 4");
    static constexpr int XS = NX > 4 ? 2 : 1;  // state slots in use
    // fp64 operand fields, [field][64 lanes]
    enum { F_MF0 = 0, F_MF1, F_MF2, F_MB0, F_MB1, F_MB2, F_FD0, F_FD1, F_APF0, F_APF1, F_BPF, NF };
    static constexpr int NROW = NX + NU;
    // fp32 pack: `nk` knots of [lo(NROW) hi(NROW)] (state rows of knot k, input rows of knot k; nk = 1 when the bounds do
    // not depend on the knot, AdmmParams::bounds_stride = 0), then Qd[NX] Rd[NU], then -inf, +inf pads
    static constexpr int bounds_len(int nk) { return 2 * NROW * nk + NROW + 2; }
    // LDS state, per position (a knot's worth of the 16 instances): [A1: NROW rows][A3: NROW rows][A2: rows that lie in
    // a cone] x 16 instances, + one zero pad cell per lane for the slots a lane does not own
    static constexpr int pos_len(int cone_rows) { return 16 * (2 * NROW + cone_rows); }
    static constexpr int PAD_LEN = 64 + 16 * NROW;  // a lane's pad cell and its A3 twin (A3_DISP further on)
    // linear rows: per-lane coefficients [mlx][2 slots][64] + [mlu][64], then b and 1 / |a|^2 of every row (even count)
    static constexpr int lin_len(int mlx, int mlu) { return (2 * mlx + mlu) * 64 + ((2 * (mlx + mlu) + 1) & ~1); }
    // (`set_rows` = the rows that lie in a cone + the rows of the sides that have linear rows: duals beyond the box set's)
    static constexpr size_t lds_bytes(int N, int nk, int set_rows, int mlx = 0, int mlu = 0) {
        return sizeof(float) * ((size_t)pos_len(set_rows) * (N - 1) + PAD_LEN + ((bounds_len(nk) + 1) & ~1) + (((size_t)NROW * N + 2) & ~(size_t)1) +
                               256 + 200 + (mlx + mlu > 0 ? lin_len(mlx, mlu) : 0)) + sizeof(double) * 8;   // ... + hand-over ring + residual / flag exchange
    }
    // HBM scratch per wavefront (floats): cone slack (and behind it the linear sets' slack) of the iteration before a
    // check, [pos 0..N-1][slot][lane]
    static constexpr size_t scratch_floats(int N) { return (size_t)N * 3 * 64 * 2; }
};

// sum over the four lanes (16 apart) of an instance, on the VALU: v_permlane16_swap exchanges the odd 16-lane rows of
// its first operand with the even rows of its second, v_permlane32_swap the upper 32 lanes with the lower 32, so with
// both operands equal the two results add up to the pair sums in every lane (no LDS round trip as with ds_bpermute)
__device__ __forceinline__ float mfc_inst_sum(float v) {
    // (inline asm: through __builtin_amdgcn_permlane16_swap the compiler of this image loses the second result when
    // both are consumed by one add — it emits v_add v0, v0, v0 — experiments/permlane_probe.hip.  The s_nop covers the
    // VALU-write -> permlane-read hazard the compiler would otherwise pad for.)
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;
    b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// CX / CU: the number of second-order cones on the state / input side (0, 1 or 2; bindings.cpp:453-490 takes cone LISTS).
// The cones of a side share ONE slack and ONE dual array (admm.cpp:57-64 projects each cone's block of `x + gc` into its
// block of the slack; rows in no cone pass through), so a second cone is a second pair of instance sums over another set
// of row weights, applied to the same two values per lane.  The count is a compile-time constant: the knot loops stay
// straight-line code the scheduler can interleave with the matrix products.  More cones per side stay on the stream kernel.
// (A second cone or linear rows cost LDS, and LDS is what bounds this kernel: each tile's chain wavefront is busy a
// fraction of the time, so throughput is tiles per CU.  Rocket, N = 30: 24 floats per knot -> three tiles per CU, 27 or 30
// -> two: 7.9 / 8.6 ms against 5.2, no faster than the stream kernel on the same problems — see Solver::select_kernel.)
template <int NX, int NU, int REFS, int CX, int CU, bool BV, bool LIN = false>
__global__ __launch_bounds__(192) void admm_mfmac_kernel(const AdmmParams P) {
    using S = ConeShape<NX, NU>;
    constexpr int XS = S::XS, NROW = S::NROW;
    constexpr bool EXT = CX + CU > 0;
    static_assert(CX >= 0 && CX <= 2 && CU >= 0 && CU <= 2, "at most two cones per side");
    extern __shared__ __align__(16) unsigned char s_raw_c[];
    const int N = P.N;
    constexpr int ncx = CX, ncu = CU;
    // rows that lie in some cone get a cone dual of their own (A2); every other row's stays identically zero
    auto in_cone = [&](int rho) -> bool {
        if (rho < NX) {
            for (int c = 0; c < ncx; ++c)
                if (rho >= P.Acx[c] && rho < P.Acx[c] + P.qcx[c]) return true;
        } else {
            for (int c = 0; c < ncu; ++c)
                if (rho - NX >= P.Acu[c] && rho - NX < P.Acu[c] + P.qcu[c]) return true;
        }
        return false;
    };
    int cone_rows = 0;
    for (int rho = 0; rho < NROW; ++rho) cone_rows += in_cone(rho) ? 1 : 0;
    const int mlx = LIN ? P.mlx : 0, mlu = LIN ? P.mlu : 0;   // linear rows per side
    const int lin_rows = (mlx > 0 ? NX : 0) + (mlu > 0 ? NU : 0);
    const int PLEN = S::pos_len(cone_rows + lin_rows);
    const int nk = BV ? N : 1;   // BV: the bounds depend on the knot (per-knot pack in LDS), else one knot's worth in registers
    float *s_state = reinterpret_cast<float *>(s_raw_c);
    float *s_pad = s_state + (size_t)PLEN * (N - 1);          // S::PAD_LEN zeros: what lanes without a row read and write
    float *s_bnd = s_pad + S::PAD_LEN;
    float *s_ref = s_bnd + ((S::bounds_len(nk) + 1) & ~1);    // [N][NROW] and one zero cell behind (even offset: fp64 cells follow)
    double *s_pterm = reinterpret_cast<double *>(s_ref + (((size_t)NROW * N + 2) & ~(size_t)1));
    float *s_ring = reinterpret_cast<float *>(s_pterm + 8);   // pri_u[64] dua_u[64] (wave 2 -> 0)
    float *s_xchg = s_ring + 256;                             // pri_x[64] dua_x[64] (wave 1 -> 0), conv[64], any_left (wave 0 -> 1, 2), [196] step counter
    float *s_lin = s_xchg + 200;                              // LIN: [mlx][2][64] | [mlu][64] | b, 1 / |a|^2 per row (state rows first)
    float *s_lb = s_lin + (2 * mlx + mlu) * 64;
    __shared__ uint4 s_cmask[8 * 4];  // [cone][lane group]: x head bits, x axis bits, u head bit, u axis bit (bit = slot)

    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, g = l >> 4, j = l & 15;
    // Persistent workgroups: a workgroup takes 16-instance tiles off a global counter until none is left (the launch
    // has at most as many workgroups as fit on the chip at once).  The hardware dispatcher hands workgroups to the XCDs
    // in strict rotation, so with several rounds of long-running workgroups a slot freed on one XCD stays empty while
    // another XCD is still full (admm_mfmar.hip.h, scripts/mfmar_timeline.py).
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
            if (i == NROW * N) {
                s_ref[i] = 0.f;
                continue;
            }
            if (r < NX) v = -(P.xref[k * NX + r] * P.bounds[2 * NROW * nk + r]);
            else if (k < N - 1) v = -(P.uref[k * NU + (r - NX)] * P.bounds[2 * NROW * nk + r]);
            s_ref[i] = v;
        }
    }
    const double *gc64 = reinterpret_cast<const double *>(P.coef);
    double cf[S::NF];
#pragma unroll
    for (int f = 0; f < S::NF; ++f) cf[f] = gc64[f * 64 + l];
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
        if (tid < 32) {
            const int c = tid >> 2, gg = tid & 3;
            unsigned hx = 0u, ax = 0u, hu = 0u, au = 0u;
            if (c < ncx)
                for (int sl = 0; sl < 2; ++sl) {
                    const int row = 4 * sl + gg;
                    if (row < NX && row >= P.Acx[c] && row < P.Acx[c] + P.qcx[c] - 1) hx |= 1u << sl;
                    if (row < NX && row == P.Acx[c] + P.qcx[c] - 1) ax |= 1u << sl;
                }
            if (c < ncu) {
                if (gg < NU && gg >= P.Acu[c] && gg < P.Acu[c] + P.qcu[c] - 1) hu = 1u;
                if (gg < NU && gg == P.Acu[c] + P.qcu[c] - 1) au = 1u;
            }
            s_cmask[tid] = make_uint4(hx, ax, hu, au);
        }
    }
    if constexpr (LIN) {
        // AdmmParams::lin: [mlx][nx] rows | b | |a|^2 | [mlu][nu] rows | b | |a|^2 -> every lane's own coefficients per slot
        const float *gAx = P.lin, *gbx = gAx + mlx * NX, *gn2x = gbx + mlx;
        const float *gAu = gn2x + mlx, *gbu = gAu + mlu * NU, *gn2u = gbu + mlu;
        for (int i = tid; i < 2 * mlx * 64; i += 192) {
            const int r = i >> 7, sl = (i >> 6) & 1, row = 4 * sl + ((i & 63) >> 4);
            s_lin[i] = row < NX ? gAx[r * NX + row] : 0.f;
        }
        for (int i = tid; i < mlu * 64; i += 192) {
            const int r = i >> 6, row = (i & 63) >> 4;
            s_lin[2 * mlx * 64 + i] = row < NU ? gAu[r * NU + row] : 0.f;
        }
        for (int i = tid; i < mlx + mlu; i += 192) {
            s_lb[2 * i] = i < mlx ? gbx[i] : gbu[i - mlx];
            s_lb[2 * i + 1] = i < mlx ? gn2x[i] : gn2u[i - mlx];
        }
    }
    __syncthreads();

    constexpr bool soc_x = CX > 0, soc_u = CU > 0;
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
    const int PAD_LO = 2 * NROW * nk + NROW, PAD_HI = PAD_LO + 1;
    if (tid == 0) {
        s_bnd[PAD_LO] = -__builtin_inff();
        s_bnd[PAD_HI] = __builtin_inff();
    }
    __syncthreads();
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
    auto lo_of = [&](int k, int sl) -> float {
        if constexpr (BV) return s_bnd[k * bs3[sl] + bl[sl]];
        else return lo_c[sl];
    };
    auto hi_of = [&](int k, int sl) -> float {
        if constexpr (BV) return s_bnd[k * bs3[sl] + bh[sl]];
        else return hi_c[sl];
    };

    // ---- LDS state addressing ----
    // Position block: [A1: NROW rows][A3: NROW rows][A2: cone rows] x 16 instances.  With slot rows g, 4 + g, NX + g the
    // element of (row, instance j) sits at row * 16 + j = (row base of the slot) * 16 + l: ONE per-lane address per slot
    // serves A1 and A3 (compile-time displacement between them); lanes that do not own the slot's row point at the pad
    // (stride 0), which holds exact zeros for the whole solve: a missing row's values are zeros that stay zeros.
    constexpr int A3_DISP = NROW * 16;                         // floats from A1 to A3 of the same row
    typedef float __attribute__((address_space(3))) lds_f;
    lds_f *const sbase = (lds_f *)s_state;
    const int rbase[3] = {0, 64, NX * 16};                     // slot row base * 16 (slot 1: rows 4 + g)
    const bool okr[3] = {ok0, ok1, ok2};
    lds_f *a_ptr[3], *c_ptr[3], *l_ptr[3];                     // A1 of position 0 (A3 = + A3_DISP), A2, and the linear set's dual
    int a_str[3], c_str[3], l_str[3];                          // floats per position: PLEN, or 0 for the pad
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        const int rho_ = sl == 0 ? row0 : (sl == 1 ? row1 : NX + row2);
        int crow = 0;
        for (int r2 = 0; r2 < rho_ && r2 < NROW; ++r2) crow += in_cone(r2) ? 1 : 0;
        const bool cone = okr[sl] && in_cone(rho_);
        a_ptr[sl] = sbase + (okr[sl] ? rbase[sl] + l : PLEN * (N - 1) + l);
        a_str[sl] = okr[sl] ? PLEN : 0;
        c_ptr[sl] = sbase + (cone ? (2 * NROW + crow) * 16 + j : PLEN * (N - 1) + l);
        c_str[sl] = cone ? PLEN : 0;
        // the linear set's dual: every row of a side that has linear rows (state rows first)
        const bool lin = okr[sl] && (sl < 2 ? mlx > 0 : mlu > 0);
        const int lrow = sl < 2 ? rho_ : (mlx > 0 ? NX : 0) + row2;
        l_ptr[sl] = sbase + (lin ? (2 * NROW + cone_rows + lrow) * 16 + j : PLEN * (N - 1) + l);
        l_str[sl] = lin ? PLEN : 0;
    }
    // the affine term rides in the products: K index 11 (slot 2 of lane group 3, a row no shape uses: nu <= 3 there, else
    // it is added on the VALU) carries the constant 1 and the operand columns f / APf, BPf
    constexpr bool ONE_COL = NU <= 3;
    const bool one_lane = ONE_COL && g == 3;
    // reference pack [knot][row]: rows the lane does not own read the zero cell behind it
    const int rf_off[3] = {ok0 ? row0 : NROW * N, ok1 ? row1 : NROW * N, ok2 ? NX + row2 : NROW * N};
    const int rf_str[3] = {ok0 ? NROW : 0, ok1 ? NROW : 0, ok2 ? NROW : 0};
    (void)rf_off, (void)rf_str;
    // HBM scratch of this wavefront: cone slack kept around a check
    float *scr = P.scratch + l;                                // (+ the tile's block, set in the tile loop)
    auto SCR = [&](int pos, int sl) -> float & { return scr[((size_t)pos * 3 + sl) * 64]; };
    auto SCRL = [&](int pos, int sl) -> float & { return scr[((size_t)(N + pos) * 3 + sl) * 64]; };   // the linear sets' slack

    double x0r[2] = {0.0, 0.0};
    float g0[2] = {0.f, 0.f}, gc0[2] = {0.f, 0.f}, gl0[2] = {0.f, 0.f};  // duals of knot 0 (state side)

    int it = 0, conv = 0;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    const int ct = P.check_termination;
    const bool can_converge = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f;
    const int last_check_it = ct > 0 ? (P.max_iter / ct) * ct : 0;

    // membership of this lane's slots in each of the side's cones, as 0 / 1 weights, and mu, 1 / mu (registers)
    constexpr int CXA = CX > 0 ? CX : 1, CUA = CU > 0 ? CU : 1;
    float hxw[CXA][2], axw[CXA][2], huw[CUA], auw[CUA], mux[CXA], muu[CUA], rmux[CXA], rmuu[CUA];
#pragma unroll
    for (int c = 0; c < CXA; ++c) hxw[c][0] = hxw[c][1] = axw[c][0] = axw[c][1] = 0.f, mux[c] = rmux[c] = 1.f;
#pragma unroll
    for (int c = 0; c < CUA; ++c) huw[c] = auw[c] = 0.f, muu[c] = rmuu[c] = 1.f;
    if constexpr (EXT) {
#pragma unroll
        for (int c = 0; c < CX; ++c) {
            const uint4 mk = s_cmask[c * 4 + g];
            mux[c] = P.cx[c], rmux[c] = 1.f / P.cx[c];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) hxw[c][sl] = (float)((mk.x >> sl) & 1u), axw[c][sl] = (float)((mk.y >> sl) & 1u);
        }
#pragma unroll
        for (int c = 0; c < CU; ++c) {
            const uint4 mk = s_cmask[c * 4 + g];
            muu[c] = P.cu[c], rmuu[c] = 1.f / P.cu[c];
            huw[c] = (float)(mk.z & 1u), auw[c] = (float)(mk.w & 1u);
        }
    }
    auto cone_apply = [&](float v, float hw, float aw, float sc, float ax_new) -> float {
        v = hw != 0.f ? v * sc : v;
        return aw != 0.f ? ax_new : v;
    };
    // (the cones of a side own disjoint rows: every cone's sums are taken from the values as they came, so the cones'
    // dependency chains — two instance sums, sqrt, rcp, selects: ~150 cycles each for a lone wavefront — are independent
    // and interleave; applied one after the other they would not be, to the compiler, which cannot see the rows are disjoint)
    auto project_x = [&](float (&v)[2]) {
        float sc[CXA], ax_new[CXA];
#pragma unroll
        for (int c = 0; c < CX; ++c) {
            const float a2 = mfc_inst_sum(fmaf(hxw[c][0] * v[0], v[0], hxw[c][1] * v[1] * v[1]));
            const float axv = mfc_inst_sum(fmaf(axw[c][0], v[0], axw[c][1] * v[1]));
            cone_scale(a2, axv, mux[c], rmux[c], sc[c], ax_new[c]);
        }
#pragma unroll
        for (int c = 0; c < CX; ++c) {
            v[0] = cone_apply(v[0], hxw[c][0], axw[c][0], sc[c], ax_new[c]);
            v[1] = cone_apply(v[1], hxw[c][1], axw[c][1], sc[c], ax_new[c]);
        }
    };
    auto project_u = [&](float &v) {
        float sc[CUA], ax_new[CUA];
#pragma unroll
        for (int c = 0; c < CU; ++c) {
            const float a2 = mfc_inst_sum(huw[c] * v * v), axv = mfc_inst_sum(auw[c] * v);
            cone_scale(a2, axv, muu[c], rmuu[c], sc[c], ax_new[c]);
        }
#pragma unroll
        for (int c = 0; c < CU; ++c) v = cone_apply(v, huw[c], auw[c], sc[c], ax_new[c]);
    };

    // z <- projection onto {a_k . z <= b_k}, one row after the other (the solver's order); the dot product is summed over
    // the instance's lanes, the correction is a select (the branch would diverge between instances)
    auto halfspaces_x = [&](float (&v)[2]) {
        for (int r = 0; r < mlx; ++r) {
            const float c0 = s_lin[(2 * r) * 64 + l], c1 = s_lin[(2 * r + 1) * 64 + l], bb = s_lb[2 * r], n2 = s_lb[2 * r + 1];
            const float dot = mfc_inst_sum(fmaf(c0, v[0], c1 * v[1]));
            const float tt = (dot > bb && n2 > 0.f) ? (dot - bb) / n2 : 0.f;
            v[0] -= tt * c0, v[1] -= tt * c1;
        }
    };
    auto halfspaces_u = [&](float &v) {
        for (int r = 0; r < mlu; ++r) {
            const float c0 = s_lin[(2 * mlx + r) * 64 + l], bb = s_lb[2 * (mlx + r)], n2 = s_lb[2 * (mlx + r) + 1];
            const float dot = mfc_inst_sum(c0 * v);
            const float tt = (dot > bb && n2 > 0.f) ? (dot - bb) / n2 : 0.f;
            v -= tt * c0;
        }
    };

    // ---- three wavefronts per 16 instances ----
    // A CU's LDS holds two tiles' state at N = 50, so its four SIMDs would carry two wavefronts, each of them stalled
    // on its own recurrence most of the time.  The slack / dual work of a knot is independent of the rollout that feeds
    // it, so a tile gets three wavefronts (on different SIMDs):
    //   wave 0 ("chain"): the matrix products of both sweeps and nothing else on the forward one — each step is
    //                     products -> x_{k+1}, u_k -> LDS -> next products, ~3 x 64 matrix-core cycles;
    //   wave 1 ("state"): the state rows' box and cone sets of the forward sweep, one or more knots behind wave 0;
    //   wave 2 ("input"): the input rows' sets, likewise.
    // Hand-over: wave 0 stores x_{k+1} / u_k (fp32) into the A3 cells of position k — the cells the consumer overwrites
    // with its fused slack - dual value, whose previous content (s of the last iteration; t_k, already fed to the
    // products) is dead — then bumps a monotone step counter in LDS.  LDS executes a wavefront's accesses in order, so
    // a consumer that has seen the counter reach its step finds the values (volatile accesses keep the compiler's order);
    // wave 0 never waits for the others inside a sweep.  One workgroup barrier ends the forward sweep (the backward
    // sweep, wave 0's alone, reads what waves 1 and 2 wrote); residual maxima and the per-instance convergence flags
    // cross through LDS on the iterations that check.
    typedef volatile float __attribute__((address_space(3))) lds_vf;
    typedef volatile int __attribute__((address_space(3))) lds_vi;
    lds_vi *const s_step = (lds_vi *)reinterpret_cast<int *>(s_xchg + 196);
    int any_left = 1;                                          // some instance of the tile still iterates (all waves agree)
#ifdef TMPC_MFMAC_PROBE
    // timing probe (scripts/mfmac_cycles.py; results are NOT a solution): bit 0 no backward sweep, 1 no state work,
    // 2 no input work, 3 report s_memtime deltas per knot step in place of the residuals
    const int dbg = P.mpc_steps;
    long long T_fwd = 0, T_bar = 0, T_bwd = 0, T_w1 = 0;
#define TMPC_PROBE(x) x
#else
    constexpr int dbg = 0;
#define TMPC_PROBE(x)
#endif
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
    g0[0] = g0[1] = gc0[0] = gc0[1] = gl0[0] = gl0[1] = 0.f;
    it = 0, conv = 0, any_left = 1;
    res0 = res1 = res2 = res3 = 0.f;
    // cold start = the zero workspace tiny_setup leaves (tiny_api.cpp:73-88)
    for (int i = tid; i < PLEN * (N - 1) + S::PAD_LEN; i += 192) s_state[i] = 0.f;
    if (tid == 0) *s_step = 0;                                 // the hand-over step counter
    __syncthreads();
    for (int i = 0; i < P.max_iter; ++i) {
        const int itn = i + 1;
        const bool check = ct > 0 && itn % ct == 0;
        const bool need_res = check && (can_converge || itn == last_check_it);
        const bool check_next = ct > 0 && (itn + 1) % ct == 0 && itn < P.max_iter;
        const bool write_old = check_next && (can_converge || itn + 1 == last_check_it);   // the next iteration reads this one's slack
        const bool last = itn == P.max_iter;
        const bool write_sol = write_old || need_res || last;   // box slack -> xout / uout
        const bool read_old = need_res && itn > 1;               // the zero workspace before the first iteration
        const bool wr = active && !conv;                         // a converged instance's outputs are frozen
        const bool full = need_res || write_sol;
        const int step0 = i * (N - 1);                           // the step counter's value before this sweep
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
        lds_f *pa[3] = {a_ptr[0], a_ptr[1], a_ptr[2]};
        auto wait_step = [&](int k) {                            // until wave 0 has handed over step k of this sweep
            TMPC_PROBE(const long long tb0 = (dbg & 8) ? clock64() : 0;)
            while (*s_step < step0 + k + 1) __builtin_amdgcn_s_sleep(1);
            TMPC_PROBE(if (dbg & 8) T_w1 += clock64() - tb0;)
        };

        if (wave == 1) {
            // ================= state side of the forward sweep (admm.cpp:43-59, :65-69, :93-96) =================
            // slack / dual of both sets for this lane's two state slots at knot kn (position spos of the cone scratch)
            auto state_knot = [&](auto full_tag, const float (&xf)[2], int kn, int spos, float (&a1)[2], float (&a2)[2], float (&a4)[2], float (&sx)[2]) {
                constexpr bool FULL = decltype(full_tag)::value;
                float vn[2], vc[2];
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    vn[sl] = __builtin_amdgcn_fmed3f(xf[sl] + a1[sl], lo_of(kn, sl), hi_of(kn, sl));   // admm.cpp:52-56
                    a1[sl] = (a1[sl] + xf[sl]) - vn[sl];                                               // admm.cpp:68
                    sx[sl] = vn[sl] - a1[sl];
                }
                if constexpr (FULL) {
                    if (need_res) {
#pragma unroll
                        for (int sl = 0; sl < XS; ++sl) {
                            float old = 0.f;
                            if (read_old && active && okr[sl]) old = P.xout[b * EX + (long)kn * NX + (sl ? row1 : row0)];
                            pri_x = fmaxf(pri_x, fabsf(xf[sl] - vn[sl]));
                            dua_x = fmaxf(dua_x, fabsf(old - vn[sl]));
                        }
                    }
                    if (write_sol && wr) {
#pragma unroll
                        for (int sl = 0; sl < XS; ++sl)
                            if (okr[sl]) P.xout[b * EX + (long)kn * NX + (sl ? row1 : row0)] = vn[sl];
                    }
                }
                if constexpr (soc_x) {
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) vc[sl] = xf[sl] + a2[sl];
                    project_x(vc);
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        a2[sl] = (a2[sl] + xf[sl]) - vc[sl];
                        sx[sl] += vc[sl] - a2[sl];
                    }
                    if constexpr (FULL) {
                        if (need_res) {
#pragma unroll
                            for (int sl = 0; sl < XS; ++sl) {
                                const float old = read_old ? SCR(spos, sl) : 0.f;
                                pri_x = fmaxf(pri_x, fabsf(xf[sl] - vc[sl]));
                                dua_x = fmaxf(dua_x, fabsf(old - vc[sl]));
                            }
                        }
                        if (write_old) {
#pragma unroll
                            for (int sl = 0; sl < XS; ++sl) SCR(spos, sl) = vc[sl];
                        }
                    }
                }
                if constexpr (LIN) {
                    if (mlx > 0) {
                        float vl[2] = {xf[0] + a4[0], xf[1] + a4[1]};
                        halfspaces_x(vl);
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl) {
                            a4[sl] = (a4[sl] + xf[sl]) - vl[sl];
                            sx[sl] += vl[sl] - a4[sl];
                        }
                        if constexpr (FULL) {
                            if (need_res) {
#pragma unroll
                                for (int sl = 0; sl < XS; ++sl) {
                                    const float old = read_old ? SCRL(spos, sl) : 0.f;
                                    pri_x = fmaxf(pri_x, fabsf(xf[sl] - vl[sl]));
                                    dua_x = fmaxf(dua_x, fabsf(old - vl[sl]));
                                }
                            }
                            if (write_old) {
#pragma unroll
                                for (int sl = 0; sl < XS; ++sl) SCRL(spos, sl) = vl[sl];
                            }
                        }
                    }
                }
            };
            auto state_sweep = [&](auto full_tag) {
                {
                    const float xf0[2] = {(float)x0r[0], XS == 2 ? (float)x0r[1] : 0.f};
                    float sx0[2];
                    state_knot(full_tag, xf0, 0, N - 1, g0, gc0, gl0, sx0);   // knot 0: its fused value feeds nothing (q_0 only enters p_0)
                }
                lds_f *pc[2] = {c_ptr[0], c_ptr[1]}, *pl[2] = {l_ptr[0], l_ptr[1]};
                float nA1[2] = {*pa[0], *pa[1]}, nA2[2] = {soc_x ? *pc[0] : 0.f, soc_x ? *pc[1] : 0.f}, nA4[2] = {LIN ? *pl[0] : 0.f, LIN ? *pl[1] : 0.f};
                for (int k = 0; k < N - 1; ++k) {
                    float a1x[2] = {nA1[0], nA1[1]}, a2x[2] = {nA2[0], nA2[1]}, a4x[2] = {nA4[0], nA4[1]};
                    lds_f *const wa[2] = {pa[0], pa[1]}, *const wc[2] = {pc[0], pc[1]}, *const wl[2] = {pl[0], pl[1]};
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        pa[sl] += a_str[sl];
                        pc[sl] += c_str[sl];
                        pl[sl] += l_str[sl];
                    }
                    if (k + 1 < N - 1) {                                         // next position's duals, before the wait
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl) {
                            nA1[sl] = *pa[sl];
                            nA2[sl] = soc_x ? *pc[sl] : 0.f;
                            nA4[sl] = LIN ? *pl[sl] : 0.f;
                        }
                    }
                    wait_step(k);                                                // x_{k+1} is in the A3 cells of position k
                    const float xf[2] = {*(lds_vf *)(wa[0] + A3_DISP), XS == 2 ? *(lds_vf *)(wa[1] + A3_DISP) : 0.f};
                    float sx[2] = {0.f, 0.f};
                    if (!(dbg & 2)) state_knot(full_tag, xf, k + 1, k, a1x, a2x, a4x, sx);
#pragma unroll
                    for (int sl = 0; sl < XS; ++sl) {
                        *wa[sl] = a1x[sl];
                        wa[sl][A3_DISP] = sx[sl];
                        if constexpr (soc_x) *wc[sl] = a2x[sl];
                        if constexpr (LIN) *wl[sl] = a4x[sl];
                    }
                }
            };
            if (full) state_sweep(std::true_type{});
            else state_sweep(std::false_type{});
            if (need_res) {                                                      // hand the state side's maxima to wave 0
                s_xchg[l] = mf_inst_max(pri_x);
                s_xchg[64 + l] = mf_inst_max(dua_x);
            }
        } else if (wave == 2) {
            // ================= input side of the forward sweep (admm.cpp:43-51, :60-64, :93-96) =================
            auto input_sweep = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                lds_f *pc2 = c_ptr[2], *pl2 = l_ptr[2];
                float nA1u = *pa[2], nA2u = soc_u ? *pc2 : 0.f, nA4u = LIN ? *pl2 : 0.f;
                for (int k = 0; k < N - 1; ++k) {
                    float a1u = nA1u, a2u = nA2u, a4u = nA4u;
                    lds_f *const wa2 = pa[2], *const wc2 = pc2, *const wl2 = pl2;
                    pa[2] += a_str[2];
                    pc2 += c_str[2];
                    pl2 += l_str[2];
                    if (k + 1 < N - 1) {
                        nA1u = *pa[2];
                        nA2u = soc_u ? *pc2 : 0.f;
                        nA4u = LIN ? *pl2 : 0.f;
                    }
                    wait_step(k);                                                // u_k is in the A3 cell of position k
                    const float uf = *(lds_vf *)(wa2 + A3_DISP);
                    if (dbg & 4) continue;
                    const float zn = __builtin_amdgcn_fmed3f(uf + a1u, lo_of(k, 2), hi_of(k, 2));
                    a1u = (a1u + uf) - zn;
                    float su = zn - a1u;
                    if constexpr (FULL) {
                        if (need_res) {
                            float old = 0.f;
                            if (read_old && active && ok2) old = P.uout[b * EU + (long)k * NU + row2];
                            pri_u = fmaxf(pri_u, fabsf(uf - zn));
                            dua_u = fmaxf(dua_u, fabsf(old - zn));
                        }
                        if (write_sol && wr && ok2) P.uout[b * EU + (long)k * NU + row2] = zn;
                    }
                    if constexpr (soc_u) {
                        float zc = uf + a2u;
                        project_u(zc);
                        a2u = (a2u + uf) - zc;
                        su += zc - a2u;
                        if constexpr (FULL) {
                            if (need_res) {
                                const float old = read_old ? SCR(k, 2) : 0.f;
                                pri_u = fmaxf(pri_u, fabsf(uf - zc));
                                dua_u = fmaxf(dua_u, fabsf(old - zc));
                            }
                            if (write_old) SCR(k, 2) = zc;
                        }
                        *wc2 = a2u;
                    }
                    if constexpr (LIN) {
                        if (mlu > 0) {
                            float zl = uf + a4u;
                            halfspaces_u(zl);
                            a4u = (a4u + uf) - zl;
                            su += zl - a4u;
                            if constexpr (FULL) {
                                if (need_res) {
                                    const float old = read_old ? SCRL(k, 2) : 0.f;
                                    pri_u = fmaxf(pri_u, fabsf(uf - zl));
                                    dua_u = fmaxf(dua_u, fabsf(old - zl));
                                }
                                if (write_old) SCRL(k, 2) = zl;
                            }
                        }
                        *wl2 = a4u;
                    }
                    *wa2 = a1u;
                    wa2[A3_DISP] = su;
                }
            };
            if (full) input_sweep(std::true_type{});
            else input_sweep(std::false_type{});
            if (need_res) {
                s_ring[l] = mf_inst_max(pri_u);
                s_ring[64 + l] = mf_inst_max(dua_u);
            }
        }
        if (wave != 0) {
            __syncthreads();                                                     // end of the forward sweep
            if (need_res) {
                __syncthreads();                                                 // wave 0 has decided
                conv = (int)s_xchg[128 + l];
                any_left = (int)s_xchg[192];
            }
            it += 1;
            if (last || !any_left) break;
            continue;                                                            // the backward sweep is wave 0's
        }

        // ================= wave 0: the rollout (admm.cpp:25-35) =================
        // x+ = (A - B Kinf) x - B Quu_inv t + f,  u = -Kinf x - Quu_inv t  with t = B'p + r kept by the backward sweep
        // (d = Quu_inv t of admm.cpp:17 is never formed on its own: Quu_inv rides in the forward operand, f in the
        // column of the constant 1).  The product with t does not depend on x: the one of step k + 1 is issued behind the
        // x products of step k and runs while their result is converted and stored.
        TMPC_PROBE(const long long tf0 = (dbg & 8) ? clock64() : 0;)
        {
            auto t_product = [&](float t) -> mf_d4 {
                mf_d4 c = {0.0, 0.0, 0.0, 0.0};
                if constexpr (!ONE_COL) c[0] = cf[S::F_FD0], c[1] = cf[S::F_FD1];
                return mf_mma(cf[S::F_MF2], (double)(one_lane ? 1.f : t), c);
            };
            lds_f *ph[3] = {a_ptr[0] + A3_DISP, a_ptr[1] + A3_DISP, a_ptr[2] + A3_DISP};   // hand-over cells of position k
            mf_d4 cpre = t_product(*ph[2]);
            float t_next = N > 2 ? ph[2][a_str[2]] : 0.f;                        // t of position 1
            double xa = x0r[0], xb = x0r[1];
            for (int k = 0; k < N - 1; ++k) {
                mf_d4 c = mf_mma(cf[S::F_MF0], xa, cpre);
                if constexpr (XS == 2) c = mf_mma(cf[S::F_MF1], xb, c);
                if (k + 1 < N - 1) {
                    cpre = t_product(t_next);
                    if (k + 2 < N - 1) t_next = ph[2][2 * a_str[2]];
                }
                xa = c[0], xb = c[1];
                *(lds_vf *)ph[0] = (float)xa;                                    // x_{k+1} for the state-side wavefront
                if constexpr (XS == 2) *(lds_vf *)ph[1] = (float)xb;
                *(lds_vf *)ph[2] = (float)c[2];                                  // u_k for the input-side wavefront (t_k is spent)
                *s_step = step0 + k + 1;
#pragma unroll
                for (int sl = 0; sl < 3; ++sl) ph[sl] += a_str[sl];
            }
        }
        TMPC_PROBE(if (dbg & 8) T_fwd += clock64() - tf0;)
        TMPC_PROBE(const long long tb0 = (dbg & 8) ? clock64() : 0;)
        __syncthreads();                                                         // end of the forward sweep: s of every knot is in LDS
        TMPC_PROBE(if (dbg & 8) T_bar += clock64() - tb0;)
        it += 1;
        if (need_res) {
            const float r0 = s_xchg[l], r1 = s_xchg[64 + l] * rho, r2 = s_ring[l], r3 = s_ring[64 + l] * rho;
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
        if (last || !any_left) break;
        if (dbg & 1) continue;
        // ================= fused backward sweep (admm.cpp:75-83, :13-20), wave 0 =================
        TMPC_PROBE(const long long tq0 = (dbg & 8) ? clock64() : 0;)
        // position addresses at N - 2
        lds_f *qa[3];
        qa[0] = a_ptr[0] + (N - 2) * a_str[0] + A3_DISP;
        qa[1] = a_ptr[1] + (N - 2) * a_str[1] + A3_DISP;
        qa[2] = a_ptr[2] + (N - 2) * a_str[2] + A3_DISP;
        double p[2], r_held;
        {
            const int pos = N - 2;
            double pt0 = 0.0, pt1 = 0.0;
            if constexpr (REFS == REF_SHARED) {
                pt0 = ok0 ? s_pterm[row0] : 0.0;
                pt1 = ok1 ? s_pterm[row1] : 0.0;
            }
            p[0] = pt0 - (double)(rho * *qa[0]);                                 // admm.cpp:81-82
            p[1] = pt1 - (double)(rho * *qa[1]);
            float rr = 0.f;
            if constexpr (REFS == REF_SHARED) rr = s_ref[pos * rf_str[2] + rf_off[2]];
            r_held = (double)(rr - rho * *qa[2]);                                // admm.cpp:77-78
        }
        // A stage (position i2) produces p and t of knot i2 + 1:  p- = q + AmBKt p - Kinf' r (+ AmBKt Pinf f),
        // t = B'p + r (+ B' Pinf f), from q_{i2+1} (s of the state slots at position i2), r_{i2+1} (su of the input slot
        // one position up, held from the stage before) and the p of the stage before; the affine constants ride in the
        // column of the constant 1 (nu = 4: added here).  Only the products with p are on the recurrence: a stage's
        // accumulator start {q, r} and its product with r are formed and issued one stage ahead, behind the previous
        // stage's p products, from operands read from LDS two stages ahead.
        auto stage_operands = [&](int i2, float (&sv)[3], float (&rf)[3]) {
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                qa[sl] -= a_str[sl];
                sv[sl] = *qa[sl];
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
            if (i2 < 0) c[0] = c[1] = 0.0;                                       // q_0 enters p_0 only, which nothing reads
            if constexpr (!ONE_COL) c[0] += cf[S::F_APF0], c[1] += cf[S::F_APF1], c[2] += cf[S::F_BPF];
            return mf_mma(cf[S::F_MB2], one_lane ? 1.0 : r_in, c);               // [-Kinf^T; 0] r
        };
        float sv[3] = {0.f, 0.f, 0.f}, rf[3] = {0.f, 0.f, 0.f};
        lds_f *tw = a_ptr[2] + (N - 2) * a_str[2] + A3_DISP;                     // where t of the stage's knot goes
        if (N >= 3) stage_operands(N - 3, sv, rf);
        mf_d4 cpre = stage_start(N - 3, sv, rf, r_held);                         // (N = 2: the only stage is i2 = -1)
        r_held = (double)(rf[2] - rho * sv[2]);                                  // r of knot N - 3, for the stage after
        if (N >= 4) stage_operands(N - 4, sv, rf);
        for (int i2 = N - 3; i2 >= -1; --i2) {
            mf_d4 c = mf_mma(cf[S::F_MB0], p[0], cpre);                          // + [AmBKt; B^T] p
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
        TMPC_PROBE(if (dbg & 8) T_bwd += clock64() - tq0;)
    }
#ifdef TMPC_MFMAC_PROBE
    if (dbg & 16) {                                            // where the wavefronts run: HW_ID (simd_id bits 5:4, cu_id 11:8, se_id 15:13)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        s_xchg[wave * 64 + l] = (float)((hw >> 4) & 3u);
        if (wave == 0) s_ring[l] = (float)(((hw >> 8) & 15u) + 16u * ((hw >> 13) & 7u));
        __syncthreads();
        res0 = s_xchg[l], res1 = s_xchg[64 + l], res2 = s_xchg[128 + l], res3 = s_ring[l];
    }
    if (dbg & 8) {
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
