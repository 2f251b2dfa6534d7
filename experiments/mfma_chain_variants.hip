// Which formulation of "two products on x, result fed back as the next operand" is fastest on gfx950 (one wavefront):
// cycles (s_memtime) per step.  A: acc from another register set; B: acc copied in place first; C/D: one product only
// (C: separate acc registers, D: in place); E: two independent products + fp64 adds; F: as A with a third, independent
// product issued FIRST in the step; G: third product issued last (the kernels' order).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4 mma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <int V>
__global__ void k(double *out, long *cyc, int n) {
    const int l = threadIdx.x;
    const double a0 = 1e-3 + l * 1e-5, a1 = 2e-3, a2 = 3e-3;
    double xa = 0.5, xb = 0.25, t = 0.125;
    d4 zero = {0, 0, 0, 0};
    d4 cpre = {1e-3, 2e-3, 3e-3, 4e-3};
    long t0 = __builtin_amdgcn_s_memtime();
    for (int kk = 0; kk < n; ++kk) {
        d4 c;
        if (V == 0) { c = mma(a0, xa, cpre); c = mma(a1, xb, c); }
        if (V == 1) { c = cpre; asm volatile("" : "+v"(c)); c = mma(a0, xa, c); c = mma(a1, xb, c); }
        if (V == 2) { c = mma(a0, xa, cpre); }
        if (V == 3) { c = cpre; asm volatile("" : "+v"(c)); c = mma(a0, xa, c); }
        if (V == 4) { d4 c1 = mma(a0, xa, cpre), c2 = mma(a1, xb, zero); c = c1 + c2; }
        if (V == 5) { d4 cn = mma(a2, t, zero); c = mma(a0, xa, cpre); c = mma(a1, xb, c); cpre = cn; }
        if (V == 6) { c = mma(a0, xa, cpre); c = mma(a1, xb, c); cpre = mma(a2, t, zero); }
        xa = c[0], xb = c[1];
        t += 1e-9;
    }
    asm volatile("" : "+v"(xa));
    long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = xa + xb + cpre[0];
}
template <int V>
void run(const char *name, double *d, long *c) {
    const int n = 4096;
    long h;
    k<V><<<256, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    k<V><<<256, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-60s %.1f cycles per step\n", name, (double)h / n);
}
int main() {
    double *d; long *c;
    (void)hipMalloc(&d, 1024 * 256 * 8); (void)hipMalloc(&c, 32);
    run<0>("A  two chained products, acc from other registers", d, c);
    run<1>("B  two chained products, acc copied in place first", d, c);
    run<2>("C  one product, acc from other registers", d, c);
    run<3>("D  one product, acc copied in place", d, c);
    run<4>("E  two independent products + fp64 adds", d, c);
    run<5>("F  A + independent third product issued first", d, c);
    run<6>("G  A + independent third product issued last", d, c);
    return 0;
}
