// What one rollout step of the transposed-sets kernel (admm_mfmat.hip.h) costs in isolation on gfx950, one wavefront per
// SIMD: two chained products whose operands are the previous step's result registers, two accumulator tuples that swap
// roles, the hand-over (3 conversions + LDS stores) and the next start (conversions, fp64 FMAs, select) in their shadow.
// Variants add the pieces one at a time: cycles (s_memtime) per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float __attribute__((address_space(3))) lds_f;
__device__ __forceinline__ d4 mma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <int V>
__global__ __launch_bounds__(64) void k(double *out, long *cyc, int n) {
    __shared__ float cells[144 * 64];
    const int l = threadIdx.x;
    for (int i = l; i < 144 * 64; i += 64) cells[i] = 1e-3f * (i % 97);
    __syncthreads();
    const double a0 = 1e-3 + l * 1e-5, a1 = 2e-3, g0 = 1e-4, g1 = 2e-4, g2 = 3e-4, f0 = 1e-5, f1 = 2e-5;
    const bool pass = l >= 32;
    d4 cX = {0.5, 0.25, 0.1, 0.0}, cY = {1e-3, 2e-3, 3e-3, 0.0};
    float tv = cells[l], tb = cells[64 + l];
    lds_f *pp = (lds_f *)cells + l;
    auto step = [&](int kk, d4 &cprev, d4 &cacc) {
        if (V == 4 || V == 6) {
            cacc = mma(a0, cprev[0], cacc);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            cacc = mma(a0, cprev[0], cacc);
            cacc = mma(a1, cprev[1], cacc);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_f *p = pp + (kk & 31) * 144;
        if (V == 1 || V >= 3) {
            p[128] = (float)cprev[2];
            p[64] = (float)cprev[1];
            p[0] = (float)cprev[0];
        }
        if (V == 4 || V == 6) {
            __builtin_amdgcn_sched_barrier(0);
            cacc = mma(a1, cprev[1], cacc);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (V == 2 || V == 3 || V == 4) {
            const double tvd = (double)tv, tbd = (double)tb;
            double b0 = fma(g0, tvd, f0), b1 = fma(g1, tvd, f1), b2 = g2 * tvd;
            if (pass) b1 = tbd;
            cprev[0] = b0, cprev[1] = b1, cprev[2] = b2;
        }
        if (V == 5 || V == 6) {   // fp32 arithmetic for the start, one conversion per slot at the end
            float b0 = fmaf((float)g0, tv, (float)f0), b1 = fmaf((float)g1, tv, (float)f1), b2 = (float)g2 * tv;
            if (pass) b1 = tb;
            cprev[0] = (double)b0, cprev[1] = (double)b1, cprev[2] = (double)b2;
        }
        if (V == 7) {             // start = constants only (no conversions, no FMAs): what the writes into the tuple cost
            cprev[0] = f0, cprev[1] = f1, cprev[2] = g2;
        }
        if (V >= 3 && V != 7) {
            tv = p[2 * 144 + 128];
            tb = p[2 * 144 + 96];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    long t0 = __builtin_amdgcn_s_memtime();
    for (int kk = 0; kk < n; kk += 2) {
        step(kk, cX, cY);
        step(kk + 1, cY, cX);
    }
    asm volatile("" : "+v"(cX), "+v"(cY));
    long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = cX[0] + cY[1] + cX[2] + cY[3] + tv + tb;
}
template <int V>
void run(const char *name, double *d, long *c) {
    const int n = 4096;
    long h;
    k<V><<<1024, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    k<V><<<1024, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-78s %.1f cycles per step\n", name, (double)h / n);
}
int main() {
    double *d; long *c;
    (void)hipMalloc(&d, 1024 * 1024 * 8); (void)hipMalloc(&c, 32);
    run<0>("0  two chained products on the other tuple's results, nothing else", d, c);
    run<1>("1  + hand-over (3 cvt f64->f32, 3 LDS stores)", d, c);
    run<2>("2  + next start (2 cvt f32->f64, 3 fp64 FMA, select) written into the operand tuple", d, c);
    run<3>("3  + both, + 2 LDS loads (the kernel's step)", d, c);
    run<4>("4  as 3, hand-over between the two products", d, c);
    run<5>("5  as 3, start formed in fp32 (3 cvt f32->f64 at the end)", d, c);
    run<6>("6  as 5, hand-over between the two products", d, c);
    run<7>("7  hand-over + constants written into the operand tuple", d, c);
    return 0;
}
