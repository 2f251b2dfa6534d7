// Cost table for instructions issued behind the two chained fp64 products of a step (experiments/mfmat_step_probe.hip,
// variant 0 = 152 cycles): each variant adds NI instructions of one kind per step; cycles per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float __attribute__((address_space(3))) lds_f;
__device__ __forceinline__ d4 mma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <int V, int NI>
__global__ __launch_bounds__(64) void k(double *out, long *cyc, int n) {
    __shared__ float cells[144 * 64];
    const int l = threadIdx.x;
    for (int i = l; i < 144 * 64; i += 64) cells[i] = 1e-3f * (i % 97);
    __syncthreads();
    const double a0 = 1e-3 + l * 1e-5, a1 = 2e-3;
    d4 cX = {0.5, 0.25, 0.1, 0.0}, cY = {1e-3, 2e-3, 3e-3, 0.0};
    lds_f *pp = (lds_f *)cells + l;
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    double d[8] = {1., 2., 3., 4., 5., 6., 7., 8.};
    auto step = [&](int kk, d4 &cprev, d4 &cacc) {
        cacc = mma(a0, cprev[0], cacc);
        cacc = mma(a1, cprev[1], cacc);
        __builtin_amdgcn_sched_barrier(0);
        lds_f *p = pp + (kk & 31) * 144;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (V == 1) { f[i] = (float)d[i]; asm volatile("" : "+v"(f[i])); }                 // cvt f64 -> f32 of an unrelated register
            if (V == 2) { d[i] = (double)f[i]; asm volatile("" : "+v"(d[i])); }                 // cvt f32 -> f64
            if (V == 3) { d[i] = fma(d[i], 1.0000001, 1e-9); asm volatile("" : "+v"(d[i])); }  // fp64 FMA
            if (V == 4) { f[i] = fmaf(f[i], 1.0000001f, 1e-9f); asm volatile("" : "+v"(f[i])); } // fp32 FMA
            if (V == 5) { p[i * 16] = f[i]; }                                                   // LDS store
            if (V == 6) { f[i] = p[i * 16]; asm volatile("" : "+v"(f[i])); }                    // LDS load (+ wait)
            if (V == 7) { f[i] = (float)cprev[i % 3]; asm volatile("" : "+v"(f[i])); }          // cvt f64 -> f32 of the previous RESULT
            if (V == 8) { cprev[i % 3] = d[i]; asm volatile("" : "+v"(cprev)); }                // move into the operand tuple
            if (V == 9) { asm volatile("s_nop 0"); }
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
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + f[i];
    out[blockIdx.x * 64 + threadIdx.x] = cX[0] + cY[1] + cX[2] + cY[3] + s;
}
template <int V, int NI>
void run(const char *name, double *d, long *c) {
    const int n = 4096;
    long h;
    k<V, NI><<<1024, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    k<V, NI><<<1024, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-52s x%d  %.1f cycles per step\n", name, NI, (double)h / n);
}
int main() {
    double *d; long *c;
    (void)hipMalloc(&d, 1024 * 1024 * 8); (void)hipMalloc(&c, 32);
    run<0, 1>("nothing", d, c);
    run<1, 3>("cvt f64->f32 (unrelated)", d, c);  run<1, 6>("cvt f64->f32 (unrelated)", d, c);
    run<2, 3>("cvt f32->f64", d, c);              run<2, 6>("cvt f32->f64", d, c);
    run<3, 3>("fp64 FMA", d, c);                  run<3, 6>("fp64 FMA", d, c);
    run<4, 3>("fp32 FMA", d, c);                  run<4, 6>("fp32 FMA", d, c);  run<4, 8>("fp32 FMA", d, c);
    run<5, 3>("LDS store", d, c);                 run<5, 6>("LDS store", d, c);
    run<6, 3>("LDS load + use", d, c);            run<6, 6>("LDS load + use", d, c);
    run<7, 3>("cvt f64->f32 of the previous result", d, c);
    run<8, 3>("move into the operand tuple", d, c);
    run<9, 4>("s_nop 0", d, c);                   run<9, 8>("s_nop 0", d, c);
    return 0;
}
