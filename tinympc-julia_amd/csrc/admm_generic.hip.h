// Generic fused ADMM kernel: any (nx, nu, N) at run time.
//
// Used only for problem shapes that have no specialised quad-kernel instantiation
// (admm_quad.hip.h).  One lane per instance; the reference's 12 trajectory matrices
// (types.hpp:85-104) live in an HBM scratch block laid out [element][batch] so that
// lanes of a wavefront touch consecutive addresses.  Phases are kept separate and in
// the reference's order (admm.cpp:109-207) — this path is about coverage, not speed.
//
// ST is the type of everything the ADMM iteration STORES (the 12 trajectories, the duals and slacks of every constraint
// set, the workspace kept between solves): float for precision 0 / 1 — what every other kernel of the library carries —
// or double for precision 2 (tinympc_set_precision(s, 2)): the reference's own arithmetic end to end (types.hpp:15), for
// callers who want its digits back rather than the 1e-5 of the fp32-state kernels (ill-conditioned families miss 1e-5
// there by rounding their duals to fp32 every iteration: profiles/r03_fuzz_large.txt).  The precision-2 workspace lives in
// its own fp64 block (P.ws64, layout below); inputs x0 / references and the returned solution stay fp32 arrays.
#pragma once
#include <hip/hip_runtime.h>

#include "admm_params.h"

namespace tmpc {

// coefficient pack (column-major, elements of RT): A, B, Kinf, Pinf, Quu_inv, AmBKt;
// the fp32 diagonals Qd, Rd ride at the end of the bounds pack
struct GenericPack {
    int oA, oB, oK, oP, oQi, oAt, oF, oAPf, oBPf, len;
    __host__ __device__ GenericPack(int nx, int nu) {
        oA = 0;
        oB = oA + nx * nx;
        oK = oB + nx * nu;
        oP = oK + nu * nx;
        oQi = oP + nx * nx;
        oAt = oQi + nu * nu;
        oF = oAt + nx * nx;  // affine dynamics term and its Riccati-gradient images (zeros without fdyn)
        oAPf = oF + nx;
        oBPf = oAPf + nx;
        len = oBPf + nu;
    }
};
// bounds pack: xmin[N*nx] xmax[N*nx] umin[(N-1)*nu] umax[(N-1)*nu] Qd[nx] Rd[nu]
// scratch arrays, in units of E_x / E_u blocks of [element][batch]
//   x q v vnew g  (5 x E_x)   then   u r d z znew y  (6 x E_u); p is a running vector;
//   with cones: + vc vcnew gc (3 x E_x) and zc zcnew yc (3 x E_u); with linear inequalities: + vl vlnew gl, zl zlnew yl

__device__ __forceinline__ float gfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double gfma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ void gupmax(float &m, float v) { m = fmaxf(m, fabsf(v)); }
__device__ __forceinline__ void gupmax(double &m, double v) { m = fmax(m, fabs(v)); }

// fp64 workspace block of precision 2, instance-major like the fp32 arrays: d y z [B][EU] | g v [B][EX], then per further
// constraint set (cones, linear rows): g v [B][EX] | y z [B][EU]
struct Ws64 {
    double *sd, *sy, *sz, *sg, *sv, *sgc, *svc, *syc, *szc, *sgl, *svl, *syl, *szl;
    __host__ __device__ Ws64(double *base, long B, long EX, long EU) {
        sd = base, sy = sd + B * EU, sz = sy + B * EU, sg = sz + B * EU, sv = sg + B * EX;
        sgc = sv + B * EX, svc = sgc + B * EX, syc = svc + B * EX, szc = syc + B * EU;
        sgl = szc + B * EU, svl = sgl + B * EX, syl = svl + B * EX, szl = syl + B * EU;
    }
    __host__ __device__ static long doubles(long B, long EX, long EU, int sets) { return B * (EU + (long)sets * 2 * (EX + EU)); }
};

__device__ __forceinline__ float gmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double gmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float gmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double gmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float gsqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ double gsqrt(double a) { return sqrt(a); }

template <class RT, class ST = float>
__global__ __launch_bounds__(256) void admm_generic_kernel(const AdmmParams P) {
    constexpr bool WIDE = sizeof(ST) == 8;
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.batch) return;
    const int nx = P.nx, nu = P.nu, N = P.N;
    const long B = P.batch;
    const int EX = nx * N, EU = nu * (N - 1);
    const GenericPack pk(nx, nu);
    const RT *coef = reinterpret_cast<const RT *>(P.coef);
    const RT *cA = coef + pk.oA, *cB = coef + pk.oB, *cK = coef + pk.oK, *cP = coef + pk.oP,
             *cQi = coef + pk.oQi, *cAt = coef + pk.oAt, *cF = coef + pk.oF, *cAPf = coef + pk.oAPf,
             *cBPf = coef + pk.oBPf;
    const bool soc_x = P.ncx > 0, soc_u = P.ncu > 0, lin_x = P.mlx > 0, lin_u = P.mlu > 0;
    // linear-inequality pack: rows, right-hand sides, squared row norms, states then inputs
    const float *lAx = P.lin, *lbx = lAx + P.mlx * nx, *ln2x = lbx + P.mlx;
    const float *lAu = ln2x + P.mlx, *lbu = lAu + P.mlu * nu, *ln2u = lbu + P.mlu;
    const float *xmin = P.bounds, *xmax = P.bounds + EX, *umin = P.bounds + 2 * EX,
                *umax = P.bounds + 2 * EX + EU, *cQd = P.bounds + 2 * EX + 2 * EU,
                *cRd = P.bounds + 2 * EX + 2 * EU + nx;
    ST *sx = reinterpret_cast<ST *>(P.scratch) + b, *sq = sx + (long)EX * B, *sv = sq + (long)EX * B, *svn = sv + (long)EX * B, *sg = svn + (long)EX * B;
    ST *su = sg + (long)EX * B, *sr = su + (long)EU * B, *sd = sr + (long)EU * B,
       *sz = sd + (long)EU * B, *szn = sz + (long)EU * B, *sy = szn + (long)EU * B;
    ST *svc = sy + (long)EU * B, *svcn = svc + (long)EX * B, *sgc = svcn + (long)EX * B;
    ST *szc = sgc + (long)EX * B, *szcn = szc + (long)EU * B, *syc = szcn + (long)EU * B;
    ST *svl = syc + (long)EU * B, *svln = svl + (long)EX * B, *sgl = svln + (long)EX * B;
    ST *szl = sgl + (long)EX * B, *szln = szl + (long)EU * B, *syl = szln + (long)EU * B;
    // the workspace kept between solves: the fp32 arrays, or (precision 2) the fp64 block
    const Ws64 w64(P.ws64, B, EX, EU);
    auto WSP = [&](float *f32, double *f64) -> ST * {
        if constexpr (WIDE) return reinterpret_cast<ST *>(f64);
        else return reinterpret_cast<ST *>(f32);
    };
    ST *const Psd = WSP(P.sd, w64.sd), *const Psy = WSP(P.sy, w64.sy), *const Psz = WSP(P.sz, w64.sz), *const Psg = WSP(P.sg, w64.sg),
       *const Psv = WSP(P.sv, w64.sv), *const Psgc = WSP(P.sgc, w64.sgc), *const Psvc = WSP(P.svc, w64.svc),
       *const Psyc = WSP(P.syc, w64.syc), *const Pszc = WSP(P.szc, w64.szc), *const Psgl = WSP(P.sgl, w64.sgl),
       *const Psvl = WSP(P.svl, w64.svl), *const Psyl = WSP(P.syl, w64.syl), *const Pszl = WSP(P.szl, w64.szl);
#define AT(arr, e) arr[(long)(e)*B]
    // adaptive rho: this instance's own rho, Kinf (nu x nx), Pinf (nx x nx) instead of the family's
    const bool adaptive = P.adaptive_rho != 0;
    double *arho = P.adapt + b, *aK = arho + B, *aP = aK + (long)nu * nx * B;
    ST rho = adaptive ? (ST)arho[0] : (WIDE ? (ST)P.rho_family : (ST)P.rho);
    auto Kc = [&](int a, int j) -> RT { return adaptive ? (RT)AT(aK, a + j * nu) : cK[a + j * nu]; };
    auto Pc = [&](int r, int j) -> RT { return adaptive ? (RT)AT(aP, r + j * nx) : cP[r + j * nx]; };
    const bool warm = !P.cold_start;
    for (int e = 0; e < EX; ++e) {
        AT(sx, e) = e < nx ? (ST)P.x0[b * nx + e] : (ST)0;
        AT(sg, e) = warm ? Psg[b * EX + e] : (ST)0;
        AT(sv, e) = warm ? Psv[b * EX + e] : (ST)0;
        AT(svn, e) = (ST)0;
        AT(sq, e) = (ST)0;
        if (soc_x) {
            AT(svc, e) = warm ? Psvc[b * EX + e] : (ST)0;
            AT(sgc, e) = warm ? Psgc[b * EX + e] : (ST)0;
            AT(svcn, e) = (ST)0;
        }
        if (lin_x) {
            AT(svl, e) = warm ? Psvl[b * EX + e] : (ST)0;
            AT(sgl, e) = warm ? Psgl[b * EX + e] : (ST)0;
            AT(svln, e) = (ST)0;
        }
    }
    for (int e = 0; e < EU; ++e) {
        AT(su, e) = (ST)0;
        AT(sr, e) = (ST)0;
        AT(szn, e) = (ST)0;
        AT(sd, e) = warm ? Psd[b * EU + e] : (ST)0;
        AT(sy, e) = warm ? Psy[b * EU + e] : (ST)0;
        AT(sz, e) = warm ? Psz[b * EU + e] : (ST)0;
        if (soc_u) {
            AT(szc, e) = warm ? Pszc[b * EU + e] : (ST)0;
            AT(syc, e) = warm ? Psyc[b * EU + e] : (ST)0;
            AT(szcn, e) = (ST)0;
        }
        if (lin_u) {
            AT(szl, e) = warm ? Pszl[b * EU + e] : (ST)0;
            AT(syl, e) = warm ? Psyl[b * EU + e] : (ST)0;
            AT(szln, e) = (ST)0;
        }
    }
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
    if (warm) {
        res0 = P.res[b * 4 + 0];
        res1 = P.res[b * 4 + 1];
        res2 = P.res[b * 4 + 2];
        res3 = P.res[b * 4 + 3];
    }
    auto xref = [&](int k, int r) -> float {
        if (P.ref_mode == REF_SHARED) return P.xref[k * nx + r];
        if (P.ref_mode == REF_PER_INSTANCE) return P.xref[b * EX + k * nx + r];
        return 0.f;
    };
    auto uref = [&](int k, int r) -> float {
        if (P.ref_mode == REF_SHARED) return P.uref[k * nu + r];
        if (P.ref_mode == REF_PER_INSTANCE) return P.uref[b * EU + k * nu + r];
        return 0.f;
    };
    int it = 0, conv = 0;
    RT t[GEN_MAX_NU], xv[GEN_MAX_NX], xn[GEN_MAX_NX], uv[GEN_MAX_NU], rv[GEN_MAX_NU];
    for (int i = 0; i < P.max_iter; ++i) {
        // forward_pass — admm.cpp:25-35
        for (int j = 0; j < nx; ++j) xv[j] = (RT)AT(sx, j);
        for (int k = 0; k < N - 1; ++k) {
            for (int a = 0; a < nu; ++a) {
                RT acc = 0;
                for (int j = 0; j < nx; ++j) acc = gfma(Kc(a, j), xv[j], acc);
                uv[a] = -acc - (RT)AT(sd, k * nu + a);
                AT(su, k * nu + a) = (ST)uv[a];
            }
            for (int r = 0; r < nx; ++r) {
                RT acc = cF[r];  // + fdyn (zero unless set)
                for (int a = 0; a < nu; ++a) acc = gfma(cB[r + a * nx], uv[a], acc);
                for (int j = 0; j < nx; ++j) acc = gfma(cA[r + j * nx], xv[j], acc);
                xn[r] = acc;
                AT(sx, (k + 1) * nx + r) = (ST)acc;
            }
            for (int r = 0; r < nx; ++r) xv[r] = xn[r];
        }
        // update_slack, update_dual, update_linear_cost, residuals — admm.cpp:43-96
        ST pri_x = 0, dua_x = 0, pri_u = 0, dua_u = 0;
        for (int e = 0; e < EU; ++e) {
            const int k = e / nu, a = e % nu;
            const ST u = AT(su, e);
            ST zn = u + AT(sy, e);
            zn = gmin((ST)umax[e], gmax((ST)umin[e], zn));
            const ST yy = (AT(sy, e) + u) - zn;
            AT(sy, e) = yy;
            AT(szn, e) = zn;
            AT(sr, e) = -((ST)uref(k, a) * (ST)cRd[a]) - rho * (zn - yy);
            gupmax(pri_u, u - zn);
            gupmax(dua_u, AT(sz, e) - zn);
        }
        if (soc_u) {  // UNPINNED: cone slack zc = proj(u + yc), dual yc, extra -rho (zc - yc) in r
            for (int k = 0; k < N - 1; ++k) {
                for (int a = 0; a < nu; ++a) AT(szcn, k * nu + a) = AT(su, k * nu + a) + AT(syc, k * nu + a);
                for (int c = 0; c < P.ncu; ++c) {
                    const int s0 = k * nu + P.Acu[c], qd = P.qcu[c];
                    const ST mu = (ST)P.cu[c];
                    ST a2 = 0;
                    for (int j = 0; j < qd - 1; ++j) a2 = gfma(AT(szcn, s0 + j), AT(szcn, s0 + j), a2);
                    const ST an = gsqrt(a2), u0 = AT(szcn, s0 + qd - 1) * mu;
                    if (an <= -u0) {
                        for (int j = 0; j < qd; ++j) AT(szcn, s0 + j) = (ST)0;
                    } else if (an > u0) {
                        const ST sc = (ST)0.5 * ((ST)1 + u0 / an);
                        for (int j = 0; j < qd - 1; ++j) AT(szcn, s0 + j) *= sc;
                        AT(szcn, s0 + qd - 1) = sc * (an / mu);
                    }
                }
                for (int a = 0; a < nu; ++a) {
                    const int e = k * nu + a;
                    const ST u = AT(su, e), zc = AT(szcn, e);
                    const ST yy = (AT(syc, e) + u) - zc;
                    AT(syc, e) = yy;
                    AT(sr, e) -= rho * (zc - yy);
                    gupmax(pri_u, u - zc);
                    gupmax(dua_u, AT(szc, e) - zc);
                }
            }
        }
        if (lin_u) {  // UNPINNED: slack zl = (u + yl) projected row by row onto {a.z <= b}, dual yl, -rho (zl - yl) in r
            for (int k = 0; k < N - 1; ++k) {
                for (int a = 0; a < nu; ++a) AT(szln, k * nu + a) = AT(su, k * nu + a) + AT(syl, k * nu + a);
                for (int c = 0; c < P.mlu; ++c) {
                    ST dot = 0;
                    for (int j = 0; j < nu; ++j) dot = gfma((ST)lAu[c * nu + j], AT(szln, k * nu + j), dot);
                    if (dot > (ST)lbu[c]) {
                        const ST tt = (dot - (ST)lbu[c]) / (ST)ln2u[c];
                        for (int j = 0; j < nu; ++j) AT(szln, k * nu + j) -= tt * (ST)lAu[c * nu + j];
                    }
                }
                for (int a = 0; a < nu; ++a) {
                    const int e = k * nu + a;
                    const ST u = AT(su, e), zl = AT(szln, e);
                    const ST yy = (AT(syl, e) + u) - zl;
                    AT(syl, e) = yy;
                    AT(sr, e) -= rho * (zl - yy);
                    gupmax(pri_u, u - zl);
                    gupmax(dua_u, AT(szl, e) - zl);
                }
            }
        }
        for (int e = 0; e < EX; ++e) {
            const int k = e / nx, r = e % nx;
            const ST x = AT(sx, e);
            ST vn = x + AT(sg, e);
            vn = gmin((ST)xmax[e], gmax((ST)xmin[e], vn));
            const ST gg = (AT(sg, e) + x) - vn;
            AT(sg, e) = gg;
            AT(svn, e) = vn;
            AT(sq, e) = -((ST)xref(k, r) * (ST)cQd[r]) - rho * (vn - gg);
            gupmax(pri_x, x - vn);
            gupmax(dua_x, AT(sv, e) - vn);
        }
        if (soc_x) {  // UNPINNED: state cones, same construction
            for (int k = 0; k < N; ++k) {
                for (int r = 0; r < nx; ++r) AT(svcn, k * nx + r) = AT(sx, k * nx + r) + AT(sgc, k * nx + r);
                for (int c = 0; c < P.ncx; ++c) {
                    const int s0 = k * nx + P.Acx[c], qd = P.qcx[c];
                    const ST mu = (ST)P.cx[c];
                    ST a2 = 0;
                    for (int j = 0; j < qd - 1; ++j) a2 = gfma(AT(svcn, s0 + j), AT(svcn, s0 + j), a2);
                    const ST an = gsqrt(a2), u0 = AT(svcn, s0 + qd - 1) * mu;
                    if (an <= -u0) {
                        for (int j = 0; j < qd; ++j) AT(svcn, s0 + j) = (ST)0;
                    } else if (an > u0) {
                        const ST sc = (ST)0.5 * ((ST)1 + u0 / an);
                        for (int j = 0; j < qd - 1; ++j) AT(svcn, s0 + j) *= sc;
                        AT(svcn, s0 + qd - 1) = sc * (an / mu);
                    }
                }
                for (int r = 0; r < nx; ++r) {
                    const int e = k * nx + r;
                    const ST x = AT(sx, e), vc = AT(svcn, e);
                    const ST gg = (AT(sgc, e) + x) - vc;
                    AT(sgc, e) = gg;
                    AT(sq, e) -= rho * (vc - gg);
                    gupmax(pri_x, x - vc);
                    gupmax(dua_x, AT(svc, e) - vc);
                }
            }
        }
        if (lin_x) {  // UNPINNED: state side, same construction
            for (int k = 0; k < N; ++k) {
                for (int r = 0; r < nx; ++r) AT(svln, k * nx + r) = AT(sx, k * nx + r) + AT(sgl, k * nx + r);
                for (int c = 0; c < P.mlx; ++c) {
                    ST dot = 0;
                    for (int j = 0; j < nx; ++j) dot = gfma((ST)lAx[c * nx + j], AT(svln, k * nx + j), dot);
                    if (dot > (ST)lbx[c]) {
                        const ST tt = (dot - (ST)lbx[c]) / (ST)ln2x[c];
                        for (int j = 0; j < nx; ++j) AT(svln, k * nx + j) -= tt * (ST)lAx[c * nx + j];
                    }
                }
                for (int r = 0; r < nx; ++r) {
                    const int e = k * nx + r;
                    const ST x = AT(sx, e), vl = AT(svln, e);
                    const ST gg = (AT(sgl, e) + x) - vl;
                    AT(sgl, e) = gg;
                    AT(sq, e) -= rho * (vl - gg);
                    gupmax(pri_x, x - vl);
                    gupmax(dua_x, AT(svl, e) - vl);
                }
            }
        }
        for (int r = 0; r < nx; ++r) {
            const int e = (N - 1) * nx + r;
            RT acc = 0;
            for (int j = 0; j < nx; ++j) acc = gfma(Pc(j, r), (RT)xref(N - 1, j), acc);
            ST tail = rho * (AT(svn, e) - AT(sg, e));
            if (soc_x) tail += rho * (AT(svcn, e) - AT(sgc, e));
            if (lin_x) tail += rho * (AT(svln, e) - AT(sgl, e));
            xn[r] = -acc - (RT)tail;  // p_{N-1}
        }
        it += 1;
        // adaptive rho — admm.cpp:147-174 with rho_benchmark.cpp:44-213, the sparse products written out:
        //   rows i = 0..N-2: u_i against znew_i, A x_i + B u_i (+ fdyn) - x_{i+1} against vnew_{i+1};
        //   P = blkdiag(Q~, R~, .., Pinf), q = [Q~ x_i; R~ u_i] (zero reference), A' y with y = [y_i; g_{i+1}]
        if (adaptive && i > 0 && i % 5 == 0) {
            RT pri = 0, axm = 0, zm = 0, dres = 0, pxm = 0, atym = 0, qm = 0;
            for (int k = 0; k < N - 1; ++k) {
                for (int a = 0; a < nu; ++a) {
                    const RT u = (RT)AT(su, k * nu + a), zn = (RT)AT(szn, k * nu + a);
                    gupmax(pri, u - zn);
                    gupmax(axm, u);
                    gupmax(zm, zn);
                }
                for (int r = 0; r < nx; ++r) {
                    RT acc = cF[r];
                    for (int j = 0; j < nx; ++j) acc = gfma(cA[r + j * nx], (RT)AT(sx, k * nx + j), acc);
                    for (int a = 0; a < nu; ++a) acc = gfma(cB[r + a * nx], (RT)AT(su, k * nu + a), acc);
                    acc -= (RT)AT(sx, (k + 1) * nx + r);
                    const RT vn = (RT)AT(svn, (k + 1) * nx + r);
                    gupmax(pri, acc - vn);
                    gupmax(axm, acc);
                    gupmax(zm, vn);
                }
            }
            for (int k = 0; k < N; ++k) {
                for (int r = 0; r < nx; ++r) {
                    const RT x = (RT)AT(sx, k * nx + r), qv = (RT)cQd[r] * x;
                    RT px = qv, aty = 0;
                    if (k == N - 1) {
                        px = 0;
                        for (int j = 0; j < nx; ++j) px = gfma(Pc(r, j), (RT)AT(sx, k * nx + j), px);
                    } else {
                        for (int j = 0; j < nx; ++j) aty = gfma(cA[j + r * nx], (RT)AT(sg, (k + 1) * nx + j), aty);
                    }
                    if (k >= 1) aty -= (RT)AT(sg, k * nx + r);
                    gupmax(dres, px + qv + aty);
                    gupmax(pxm, px);
                    gupmax(atym, aty);
                    gupmax(qm, qv);
                }
                if (k < N - 1)
                    for (int a = 0; a < nu; ++a) {
                        const RT px = (RT)cRd[a] * (RT)AT(su, k * nu + a);
                        RT aty = (RT)AT(sy, k * nu + a);
                        for (int j = 0; j < nx; ++j) aty = gfma(cB[j + a * nx], (RT)AT(sg, (k + 1) * nx + j), aty);
                        gupmax(dres, px + px + aty);
                        gupmax(pxm, px);
                        gupmax(atym, aty);
                        gupmax(qm, px);
                    }
            }
            const RT eps = (RT)1e-10, prin = axm > zm ? axm : zm;
            RT duan = pxm > atym ? pxm : atym;
            duan = qm > duan ? qm : duan;
            const RT ratio = (pri / (prin + eps)) / (dres / (duan + eps) + eps);  // predict_rho, rho_benchmark.cpp:173-195
            RT nrho = (RT)arho[0] * (RT)sqrt((double)ratio);
            if (P.rho_clip) nrho = nrho < (RT)P.rho_min ? (RT)P.rho_min : (nrho > (RT)P.rho_max ? (RT)P.rho_max : nrho);
            const double delta = (double)nrho - arho[0];
            for (int e = 0; e < nu * nx; ++e) AT(aK, e) += delta * P.sens[e];
            for (int e = 0; e < nx * nx; ++e) AT(aP, e) += delta * P.sens[nu * nx + e];
            arho[0] = (double)nrho;
            rho = (ST)nrho;
        }
        // termination_condition — admm.cpp:89-107
        if (P.check_termination > 0 && it % P.check_termination == 0) {
            // (float state: the products are rounded to fp32 before the comparison, as every fp32-state kernel does; precision 2
            // compares what the reference compares: fp64 residuals against the fp64 tolerances, admm.cpp:99-103)
            const ST r1 = dua_x * rho, r3 = dua_u * rho;
            res0 = (float)pri_x;
            res1 = (float)r1;
            res2 = (float)pri_u;
            res3 = (float)r3;
            const ST tp = WIDE ? (ST)P.abs_pri_tol64 : (ST)P.abs_pri_tol, td = WIDE ? (ST)P.abs_dua_tol64 : (ST)P.abs_dua_tol;
            if (pri_x < tp && pri_u < tp && r1 < td && r3 < td) {
                conv = 1;
                break;
            }
        }
        for (int e = 0; e < EX; ++e) AT(sv, e) = AT(svn, e);
        for (int e = 0; e < EU; ++e) AT(sz, e) = AT(szn, e);
        if (soc_x)
            for (int e = 0; e < EX; ++e) AT(svc, e) = AT(svcn, e);
        if (soc_u)
            for (int e = 0; e < EU; ++e) AT(szc, e) = AT(szcn, e);
        if (lin_x)
            for (int e = 0; e < EX; ++e) AT(svl, e) = AT(svln, e);
        if (lin_u)
            for (int e = 0; e < EU; ++e) AT(szl, e) = AT(szln, e);
        // backward_pass_grad — admm.cpp:13-20
        for (int j = 0; j < nx; ++j) xv[j] = xn[j];  // running p, kept in RT
        for (int k = N - 2; k >= 0; --k) {
            for (int a = 0; a < nu; ++a) {
                rv[a] = (RT)AT(sr, k * nu + a);
                RT acc = rv[a] + cBPf[a];
                for (int j = 0; j < nx; ++j) acc = gfma(cB[j + a * nx], xv[j], acc);
                t[a] = acc;
            }
            for (int a = 0; a < nu; ++a) {
                RT acc = 0;
                for (int c = 0; c < nu; ++c) acc = gfma(cQi[a + c * nu], t[c], acc);
                AT(sd, k * nu + a) = (ST)acc;
            }
            for (int r = 0; r < nx; ++r) {
                RT ap = (RT)AT(sq, k * nx + r) + cAPf[r], kr = 0;
                for (int j = 0; j < nx; ++j) ap = gfma(cAt[r + j * nx], xv[j], ap);
                for (int a = 0; a < nu; ++a) kr = gfma(Kc(a, r), rv[a], kr);
                xn[r] = ap - kr;
            }
            for (int r = 0; r < nx; ++r) xv[r] = xn[r];
        }
    }
    for (int e = 0; e < EX; ++e) P.xout[b * EX + e] = (float)AT(svn, e);
    for (int e = 0; e < EU; ++e) P.uout[b * EU + e] = (float)AT(szn, e);
    P.iter[b] = it;
    P.solved[b] = conv;
    P.res[b * 4 + 0] = res0;
    P.res[b * 4 + 1] = res1;
    P.res[b * 4 + 2] = res2;
    P.res[b * 4 + 3] = res3;
    if (P.save_state) {
        for (int e = 0; e < EX; ++e) {
            Psg[b * EX + e] = AT(sg, e);
            Psv[b * EX + e] = AT(sv, e);
        }
        for (int e = 0; e < EU; ++e) {
            Psy[b * EU + e] = AT(sy, e);
            Psz[b * EU + e] = AT(sz, e);
            Psd[b * EU + e] = AT(sd, e);
        }
        if (soc_x)
            for (int e = 0; e < EX; ++e) {
                Psgc[b * EX + e] = AT(sgc, e);
                Psvc[b * EX + e] = AT(svc, e);
            }
        if (soc_u)
            for (int e = 0; e < EU; ++e) {
                Psyc[b * EU + e] = AT(syc, e);
                Pszc[b * EU + e] = AT(szc, e);
            }
        if (lin_x)
            for (int e = 0; e < EX; ++e) {
                Psgl[b * EX + e] = AT(sgl, e);
                Psvl[b * EX + e] = AT(svl, e);
            }
        if (lin_u)
            for (int e = 0; e < EU; ++e) {
                Psyl[b * EU + e] = AT(syl, e);
                Pszl[b * EU + e] = AT(szl, e);
            }
    }
    atomicMax(&P.gstat[0], __float_as_uint(res0));
    atomicMax(&P.gstat[1], __float_as_uint(res1));
    atomicMax(&P.gstat[2], __float_as_uint(res2));
    atomicMax(&P.gstat[3], __float_as_uint(res3));
    if (!conv) atomicAdd(&P.gstat[4], 1u);
#undef AT
}

}  // namespace tmpc
