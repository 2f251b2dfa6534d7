// Generic fused ADMM kernel: any (nx, nu, N) at run time.
//
// Used only for problem shapes that have no specialised quad-kernel instantiation
// (admm_quad.hip.h).  One lane per instance; the reference's 12 trajectory matrices
// (types.hpp:85-104) live in an HBM scratch block laid out [element][batch] so that
// lanes of a wavefront touch consecutive addresses.  Phases are kept separate and in
// the reference's order (admm.cpp:109-207) — this path is about coverage, not speed.
#pragma once
#include <hip/hip_runtime.h>

#include "admm_params.h"

namespace tmpc {

// coefficient pack (column-major fp32): A, B, Kinf, Pinf, Quu_inv, AmBKt, Qd, Rd
struct GenericPack {
    int oA, oB, oK, oP, oQi, oAt, oQd, oRd, len;
    __host__ __device__ GenericPack(int nx, int nu) {
        oA = 0;
        oB = oA + nx * nx;
        oK = oB + nx * nu;
        oP = oK + nu * nx;
        oQi = oP + nx * nx;
        oAt = oQi + nu * nu;
        oQd = oAt + nx * nx;
        oRd = oQd + nx;
        len = oRd + nu;
    }
};
// bounds pack: xmin[N*nx] xmax[N*nx] umin[(N-1)*nu] umax[(N-1)*nu]
// scratch arrays, in units of E_x / E_u blocks of [element][batch]
//   x q p v vnew g  (6 x E_x)   then   u r d z znew y  (6 x E_u)

__global__ __launch_bounds__(256) void admm_generic_kernel(const AdmmParams P) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.batch) return;
    const int nx = P.nx, nu = P.nu, N = P.N;
    const long B = P.batch;
    const int EX = nx * N, EU = nu * (N - 1);
    const GenericPack pk(nx, nu);
    const float *cA = P.coef + pk.oA, *cB = P.coef + pk.oB, *cK = P.coef + pk.oK,
                *cP = P.coef + pk.oP, *cQi = P.coef + pk.oQi, *cAt = P.coef + pk.oAt,
                *cQd = P.coef + pk.oQd, *cRd = P.coef + pk.oRd;
    const float *xmin = P.bounds, *xmax = P.bounds + EX, *umin = P.bounds + 2 * EX,
                *umax = P.bounds + 2 * EX + EU;
    float *sx = P.scratch + b, *sq = sx + (long)EX * B, *sp = sq + (long)EX * B,
          *sv = sp + (long)EX * B, *svn = sv + (long)EX * B, *sg = svn + (long)EX * B;
    float *su = sg + (long)EX * B, *sr = su + (long)EU * B, *sd = sr + (long)EU * B,
          *sz = sd + (long)EU * B, *szn = sz + (long)EU * B, *sy = szn + (long)EU * B;
#define AT(arr, e) arr[(long)(e)*B]
    const float rho = P.rho;
    const bool warm = !P.cold_start;
    for (int e = 0; e < EX; ++e) {
        AT(sx, e) = e < nx ? P.x0[b * nx + e] : 0.f;
        AT(sg, e) = warm ? P.sg[b * EX + e] : 0.f;
        AT(sv, e) = warm ? P.sv[b * EX + e] : 0.f;
        AT(svn, e) = 0.f;
        AT(sq, e) = 0.f;
        AT(sp, e) = 0.f;
    }
    for (int e = 0; e < EU; ++e) {
        AT(su, e) = 0.f;
        AT(sr, e) = 0.f;
        AT(szn, e) = 0.f;
        AT(sd, e) = warm ? P.sd[b * EU + e] : 0.f;
        AT(sy, e) = warm ? P.sy[b * EU + e] : 0.f;
        AT(sz, e) = warm ? P.sz[b * EU + e] : 0.f;
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
    float t[GEN_MAX_NU], xv[GEN_MAX_NX], uv[GEN_MAX_NU];
    for (int i = 0; i < P.max_iter; ++i) {
        // forward_pass — admm.cpp:25-35
        for (int k = 0; k < N - 1; ++k) {
            for (int j = 0; j < nx; ++j) xv[j] = AT(sx, k * nx + j);
            for (int a = 0; a < nu; ++a) {
                float acc = 0.f;
                for (int j = 0; j < nx; ++j) acc = fmaf(cK[a + j * nu], xv[j], acc);
                uv[a] = -acc - AT(sd, k * nu + a);
                AT(su, k * nu + a) = uv[a];
            }
            for (int r = 0; r < nx; ++r) {
                float ax = 0.f, bu = 0.f;
                for (int j = 0; j < nx; ++j) ax = fmaf(cA[r + j * nx], xv[j], ax);
                for (int a = 0; a < nu; ++a) bu = fmaf(cB[r + a * nx], uv[a], bu);
                AT(sx, (k + 1) * nx + r) = ax + bu;
            }
        }
        // update_slack, update_dual, update_linear_cost, residuals — admm.cpp:43-96
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
        for (int e = 0; e < EU; ++e) {
            const int k = e / nu, a = e % nu;
            const float u = AT(su, e);
            float zn = u + AT(sy, e);
            zn = fminf(umax[e], fmaxf(umin[e], zn));
            const float yy = (AT(sy, e) + u) - zn;
            AT(sy, e) = yy;
            AT(szn, e) = zn;
            AT(sr, e) = -(uref(k, a) * cRd[a]) - rho * (zn - yy);
            pri_u = fmaxf(pri_u, fabsf(u - zn));
            dua_u = fmaxf(dua_u, fabsf(AT(sz, e) - zn));
        }
        for (int e = 0; e < EX; ++e) {
            const int k = e / nx, r = e % nx;
            const float x = AT(sx, e);
            float vn = x + AT(sg, e);
            vn = fminf(xmax[e], fmaxf(xmin[e], vn));
            const float gg = (AT(sg, e) + x) - vn;
            AT(sg, e) = gg;
            AT(svn, e) = vn;
            AT(sq, e) = -(xref(k, r) * cQd[r]) - rho * (vn - gg);
            pri_x = fmaxf(pri_x, fabsf(x - vn));
            dua_x = fmaxf(dua_x, fabsf(AT(sv, e) - vn));
        }
        for (int r = 0; r < nx; ++r) {
            const int e = (N - 1) * nx + r;
            float acc = 0.f;
            for (int j = 0; j < nx; ++j) acc = fmaf(xref(N - 1, j), cP[j + r * nx], acc);
            AT(sp, e) = -acc - rho * (AT(svn, e) - AT(sg, e));
        }
        it += 1;
        // termination_condition — admm.cpp:89-107
        if (P.check_termination > 0 && it % P.check_termination == 0) {
            res0 = pri_x;
            res1 = dua_x * rho;
            res2 = pri_u;
            res3 = dua_u * rho;
            if (res0 < P.abs_pri_tol && res2 < P.abs_pri_tol && res1 < P.abs_dua_tol &&
                res3 < P.abs_dua_tol) {
                conv = 1;
                break;
            }
        }
        for (int e = 0; e < EX; ++e) AT(sv, e) = AT(svn, e);
        for (int e = 0; e < EU; ++e) AT(sz, e) = AT(szn, e);
        // backward_pass_grad — admm.cpp:13-20
        for (int k = N - 2; k >= 0; --k) {
            for (int j = 0; j < nx; ++j) xv[j] = AT(sp, (k + 1) * nx + j);
            for (int a = 0; a < nu; ++a) {
                float acc = 0.f;
                for (int j = 0; j < nx; ++j) acc = fmaf(cB[j + a * nx], xv[j], acc);
                t[a] = acc + AT(sr, k * nu + a);
            }
            for (int a = 0; a < nu; ++a) {
                float acc = 0.f;
                for (int c = 0; c < nu; ++c) acc = fmaf(cQi[a + c * nu], t[c], acc);
                AT(sd, k * nu + a) = acc;
            }
            for (int r = 0; r < nx; ++r) {
                float ap = 0.f, kr = 0.f;
                for (int j = 0; j < nx; ++j) ap = fmaf(cAt[r + j * nx], xv[j], ap);
                for (int a = 0; a < nu; ++a) kr = fmaf(cK[a + r * nu], AT(sr, k * nu + a), kr);
                AT(sp, k * nx + r) = (AT(sq, k * nx + r) + ap) - kr;
            }
        }
    }
    for (int e = 0; e < EX; ++e) P.xout[b * EX + e] = AT(svn, e);
    for (int e = 0; e < EU; ++e) P.uout[b * EU + e] = AT(szn, e);
    P.iter[b] = it;
    P.solved[b] = conv;
    P.res[b * 4 + 0] = res0;
    P.res[b * 4 + 1] = res1;
    P.res[b * 4 + 2] = res2;
    P.res[b * 4 + 3] = res3;
    if (P.save_state) {
        for (int e = 0; e < EX; ++e) {
            P.sg[b * EX + e] = AT(sg, e);
            P.sv[b * EX + e] = AT(sv, e);
        }
        for (int e = 0; e < EU; ++e) {
            P.sy[b * EU + e] = AT(sy, e);
            P.sz[b * EU + e] = AT(sz, e);
            P.sd[b * EU + e] = AT(sd, e);
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
